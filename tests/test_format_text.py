"""metaserver's output text (metaserver.cpp:472-484) from the GPU: dsm_formatter_* / dsm_format_batch_dev against the host's
snprintf loop (dsm_format_batch), byte for byte.

CPU part: the integer-only "%f" of csrc/fmt6.h against glibc's snprintf (tests/native/fmt6_check.cpp).
GPU part: random batches (exact decimal ties, +-0, negative noise, every binary exponent, ids and 64-bit frequencies), values the
device hands back to the host (2^40 and up, infinities, NaNs), and the server goldens, whose text goes through the device formatter
in every GPU test that compares tuple text (pydsm's sinks use it; DSM_TEXT_HOST=1 switches back)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integer_only_percent_f_equals_snprintf(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "fmt6_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "fmt6_check.cpp")], check=True)
    out = subprocess.run([exe, os.environ.get("DSM_FMT6_COUNT", "4000000"), "11"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok "), out
    assert int(out.split()[1]) > 3000000


def test_kept_runs_of_a_chunk_on_the_cpu(tmp_path_factory):
    """csrc/emit_runs.h (the emitter's in-place batches when a handful of a chunk's tuples are dropped): run boundaries and the offsets'
    rebasing, split over 1..17 callers in any order, against a direct restatement (tests/native/emit_runs_check.cpp)."""
    exe = str(tmp_path_factory.mktemp("native") / "emit_runs_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "emit_runs_check.cpp")], check=True)
    out = subprocess.run([exe, "4000", "7"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok 4000 cases"), out


def _batch(pydsm, ent, path_len, pairs_per, rng):
    nt = len(ent)
    pl = rng.integers(0, path_len + 1, nt).astype(np.uint32)
    path_off = np.zeros(nt + 1, np.uint32)
    np.cumsum(pl, out=path_off[1:])
    paths = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(path_off[-1])).astype(np.uint8)
    npair = rng.integers(0, pairs_per + 1, nt).astype(np.uint32)
    pair_off = np.zeros(nt + 1, np.uint32)
    np.cumsum(npair, out=pair_off[1:])
    np_ = int(pair_off[-1])
    ids = rng.integers(0, 273, np_).astype(np.uint32)
    freqs = np.where(rng.random(np_) < 0.9, rng.integers(1, 100000, np_), rng.integers(1, 2**63, np_)).astype(np.uint64)
    freqs[: min(4, np_)] = np.array([0, 9, 10, 2**64 - 1], np.uint64)[: min(4, np_)]
    b = pydsm.TupleBatch()
    b.ntuples = nt
    keep = [path_off, paths, ent, pair_off, ids, freqs]
    b.path_off = path_off.ctypes.data_as(C.POINTER(C.c_uint32))
    b.path_bytes = paths.ctypes.data if len(paths) else None
    b.entropy = ent.ctypes.data_as(C.POINTER(C.c_double))
    b.pair_off = pair_off.ctypes.data_as(C.POINTER(C.c_uint32))
    b.ids = ids.ctypes.data_as(C.POINTER(C.c_uint32))
    b.freqs = freqs.ctypes.data_as(C.POINTER(C.c_uint64))
    return b, keep


def _host_text(pydsm, b):
    t = C.c_void_p()
    n = C.c_size_t(0)
    assert pydsm.lib().dsm_format_batch(C.byref(b), C.byref(t), C.byref(n)) == 0
    s = C.string_at(t, n.value)
    pydsm.lib().dsm_free(t)
    return s


def _doubles(rng, n):
    k = n // 8
    parts = [rng.random(k) * 8.1,                                         # entropies
             rng.integers(0, 10**8, k) / 128.0,                          # k / 2^7: seven decimals ending in 5 -- exact ties
             rng.integers(0, 10**9, k) / 1024.0 * 1e-3,
             rng.random(k) * 1e-6,                                        # around the last printed digit
             rng.integers(0, 2000001, k) * 0.5e-6,                        # decimal ties that are not binary ties
             -rng.random(k) * 8.1e-9,                                     # negative noise around zero (one sample, metaserver.cpp:389)
             np.ldexp(rng.integers(0, 2**53, k).astype(np.float64), -rng.integers(0, 80, k)),  # every binary exponent
             rng.random(n - 7 * k) * 1e12]
    x = np.concatenate(parts)
    x = np.where(np.abs(x) < 2.0**40, x, 1.0)
    x[:8] = [0.0, -0.0, 5e-324, 9.9999995, 9.9999996, 0.9999995, 4.9999995e-7, -1e-10]
    rng.shuffle(x)
    return np.ascontiguousarray(x, np.float64)


@pytest.mark.gpu
def test_device_text_equals_snprintf_on_random_batches():
    """10^8 doubles by default (DSM_FORMAT_TEST_N), in batches of 12.5 M lines: the device's lines are the host's, byte for byte."""
    import pydsm
    total = int(os.environ.get("DSM_FORMAT_TEST_N", "100000000"))
    per = 12500000
    rng = np.random.default_rng(77)
    with pydsm.Formatter(0) as f:
        done = 0
        while done < total:
            n = min(per, total - done)
            # most batches: the number alone (the entropy column is what needs care); two with paths and pairs
            rich = done in (0, per)
            b, keep = _batch(pydsm, _doubles(rng, n if not rich else n // 10), 60 if rich else 0, 4 if rich else 0, rng)
            got = f.format(C.byref(b))
            want = _host_text(pydsm, b)
            assert len(got) == len(want)
            assert got == want
            done += n
    # the one-shot entry point
    b, keep = _batch(pydsm, _doubles(rng, 1000), 40, 3, rng)
    t = C.c_void_p()
    n = C.c_size_t(0)
    assert pydsm.lib().dsm_format_batch_dev(C.byref(b), 0, C.byref(t), C.byref(n)) == 0
    assert C.string_at(t, n.value) == _host_text(pydsm, b)
    pydsm.lib().dsm_free(t)


@pytest.mark.gpu
def test_values_the_device_does_not_print_go_to_the_host():
    import pydsm
    rng = np.random.default_rng(5)
    ent = _doubles(rng, 4096)
    ent[100], ent[200], ent[300], ent[400] = 2.0**40, np.inf, np.nan, -1e300
    b, keep = _batch(pydsm, ent, 20, 2, rng)
    with pydsm.Formatter(0) as f:
        assert f.format(C.byref(b)) == _host_text(pydsm, b)
        e2 = _doubles(rng, 512)
        b2, keep2 = _batch(pydsm, e2, 20, 2, rng)          # and the formatter goes on with the device afterwards
        assert f.format(C.byref(b2)) == _host_text(pydsm, b2)
        b3 = pydsm.TupleBatch()
        b3.ntuples = 0
        assert f.format(C.byref(b3)) == b""


@pytest.mark.gpu
def test_golden_server_output_through_the_device_formatter(golden):
    """The reference server's stdout for the golden sets, from mine() with the text formatted on the GPU and on the host."""
    import pydsm
    from goldenlib import server_args_to_kw
    m = golden.manifest["sets"]["toy3"]
    idx = [pydsm.Index(golden.fmi("toy3", n)) for n in m["names"]]
    for cfg in m["server_cfgs"]:
        kw = server_args_to_kw(m["server_cfgs"][cfg])
        for p in ("A", "C", "G", "T"):
            want = golden.server_out("toy3", cfg, p)
            got, _ = pydsm.mine(idx, p, fmin=m["fmin"], **kw)
            assert got == want, (cfg, p)
            lines = []
            pydsm.mine(idx, p, fmin=m["fmin"], text=False, on_batch=lambda b: lines.append(_host_text(pydsm, b)), **kw)
            assert b"".join(lines) == want, (cfg, p)
    for ix in idx:
        ix.close()
