"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/dsmhip.h declares, and refuses to work without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dsmhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(dsm_[a-z_0-9]+)\s*\(", hdr))
    names -= {"dsm_byte_sink", "dsm_tuple_sink", "dsm_allgather_fn"}
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import pydsm
    L = pydsm.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), "libdsmhip.so does not export " + s
    assert L.dsm_abi_version() == 3


def test_struct_layouts_match_header():
    import pydsm
    p = pydsm.default_params()
    assert (p.fmin, p.pmin, p.pmax, p.maxdepth, p.world_size) == (10, 2, 0, 0xFFFFFFFF, 1)
    assert p.emax == -1.0 and p.emin == 0.0
    assert C.sizeof(pydsm.Code) == 16
    assert C.sizeof(pydsm.Stats) == 22 * 8


def test_no_cpu_fallback_without_gpu(golden):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pydsm
    with pytest.raises(pydsm.DsmError) as e:
        pydsm.Index(golden.fmi("toy3", "toy-1"))
    assert e.value.code == -19  # DSM_E_NODEV
    with pytest.raises(pydsm.DsmError) as e:     # the distance matrices have no CPU path either
        pydsm.DistMat(3, maxent=[1.0])
    assert e.value.code == -19
    with pytest.raises(pydsm.DsmError):          # nor the server-side merge
        pydsm.Trie(b"")


def test_open_errors_are_reported_not_fatal(tmp_path):
    import pydsm
    with pytest.raises(pydsm.DsmError):
        pydsm.Index(str(tmp_path / "missing.fmi"))


def test_index_probe_validates_files_on_the_host(golden, tmp_path):
    """dsm_index_probe: the header checks of FMIndex::load / metaenumerate (FMIndex.cpp:305-372, metaenumerate.cpp:243-247) without a
    device -- what dsm_node --devices runs over every sample before it creates a communicator or a thread."""
    import pydsm
    good = golden.fmi("toy3", "toy-1")
    n = pydsm.probe(good)
    assert n > 0
    raw = open(good, "rb").read()
    for name, data, code in (("cut-header.fmi", raw[:1000], -5), ("cut-body.fmi", raw[:len(raw) - len(raw) // 3], -5),
                             ("version.fmi", b"\x0d" + raw[1:], -74), ("empty.fmi", b"", -5)):
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(pydsm.DsmError) as e:
            pydsm.probe(str(p))
        assert e.value.code == code, (name, e.value.code, str(e.value))
    # counts that do not add up to n (a wrapped 32-bit count): refused, as when opening
    bad = bytearray(raw)
    off = 1 + 8 + 4 + 256 * (4 if raw[0] == 14 else 8) + 8   # the code table: {count, bits, code} x 256
    csz = 4 if raw[0] < 16 else 8
    a = off + ord("A") * (csz + 8)
    bad[a] ^= 1
    p = tmp_path / "counts.fmi"
    p.write_bytes(bytes(bad))
    with pytest.raises(pydsm.DsmError) as e:
        pydsm.probe(str(p))
    assert "sum to n" in str(e.value)
    with pytest.raises(pydsm.DsmError) as e:
        pydsm.probe(str(tmp_path / "missing.fmi"))
    assert e.value.code == -2
