"""distance matrices of smtxt2entropy.
CPU: the oracle restatement reproduces the reference tool's four output files byte for byte on the committed goldens.
GPU: dsm_distmat_* (csrc/distmat.hip) against the oracle -- counts and substring totals exact, the double matrices
within 1e-9 relative (the GPU adds the same terms in a different order; the tool prints them with %f)."""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MAN = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
CASES = [(s, c) for s, cs in sorted(MAN.get("distmat", {}).items()) for c in sorted(cs)]


def _olib():
    so = os.path.join(ROOT, "oracle", "_build", "libdistmat_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), so], check=True, stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.orc_distmat_new.restype = C.c_void_p
    L.orc_distmat_new.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_uint]
    L.orc_distmat_free.argtypes = [C.c_void_p]
    L.orc_distmat_add_text.restype = C.c_long
    L.orc_distmat_add_text.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.orc_distmat_finish.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 4 + [C.c_void_p] * 5
    L.orc_distmat_steps.argtypes = [C.c_double, C.c_void_p, C.c_int]
    L.orc_distmat_options.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_free_text.argtypes = [C.c_void_p]
    return L


def case_input(setname, case):
    info = MAN["distmat"][setname][case]
    text = b"".join(gzip.open(os.path.join(GOLD, setname, "server.%s.%s.txt.gz" % (info["server_cfg"], p)), "rb").read()
                    for p in info["prefixes"])
    args = info["args"]
    minfreq = int(args[args.index("-M") + 1]) if "-M" in args else 0
    if "-m" in args:
        maxent = [float(x) for x in args[args.index("-m") + 1].split(",")]
    else:
        buf = (C.c_double * 256)()
        n = _olib().orc_distmat_steps(float(args[args.index("-e") + 1]), buf, 256)
        maxent = list(buf[:n])
    mapping, sizes = info.get("mapping"), info.get("sizes")
    smpls = max(mapping) + 1 if mapping else len(MAN["sets"][setname]["names"])
    return text, smpls, maxent, minfreq, mapping, sizes


def oracle_run(text, smpls, maxent, minfreq, mapping=None, sizes=None):
    L = _olib()
    me = (C.c_double * len(maxent))(*maxent)
    h = L.orc_distmat_new(smpls, me, len(maxent), minfreq)
    assert h
    mp = (C.c_int * len(mapping))(*mapping) if mapping else None
    sz = (C.c_double * len(sizes))(*sizes) if sizes else None
    L.orc_distmat_options(h, mp, len(mapping) if mapping else 0, sz)
    rows = L.orc_distmat_add_text(h, text, len(text))
    assert rows == text.count(b"\n")
    outs = [C.c_void_p() for _ in range(4)]
    nm = len(maxent)
    nout = np.zeros(nm, np.uint32)
    cnt = np.zeros((nm, smpls, smpls), np.uint32)
    mats = [np.zeros((nm, smpls, smpls), np.float64) for _ in range(3)]
    L.orc_distmat_finish(h, *[C.byref(o) for o in outs], nout.ctypes.data, cnt.ctypes.data, *[m.ctypes.data for m in mats])
    texts = [C.string_at(o.value) for o in outs]
    for o in outs:
        L.orc_free_text(o)
    L.orc_distmat_free(h)
    return texts, nout, cnt, mats


@pytest.mark.parametrize("setname,case", CASES)
def test_oracle_reproduces_reference_tool_output(setname, case):
    text, smpls, maxent, minfreq, mapping, sizes = case_input(setname, case)
    texts, nout, cnt, mats = oracle_run(text, smpls, maxent, minfreq, mapping, sizes)
    for kind, got in zip(("count", "log", "sqrt", "lgamma"), texts):
        want = gzip.open(os.path.join(GOLD, setname, "distmat.%s.%s.gz" % (case, kind)), "rb").read()
        assert got == want, (setname, case, kind)
    assert nout.max() <= text.count(b"\n")


@pytest.fixture(scope="module")
def pydsm_mod():
    import pydsm
    pydsm.lib()
    return pydsm


def _cmp(res, nout, cnt, mats):
    assert (res["noutput"] == nout).all()
    assert (res["count"] == cnt).all()
    for k, want in zip(("log", "sqrt", "lgamma"), mats):
        got = res[k]
        assert np.allclose(got, want, rtol=1e-9, atol=1e-6), (k, np.abs(got - want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("setname,case", CASES)
def test_gpu_distmat_matches_oracle_on_goldens(setname, case, pydsm_mod):
    text, smpls, maxent, minfreq, mapping, sizes = case_input(setname, case)
    texts, nout, cnt, mats = oracle_run(text, smpls, maxent, minfreq, mapping, sizes)
    with pydsm_mod.DistMat(smpls, maxent=maxent, minfreq=minfreq, run_to_sample=mapping, sizes=sizes) as dm:
        half = text.rfind(b"\n", 0, len(text) // 2) + 1
        dm.add_text(text[:half])          # two batches: accumulation across calls
        dm.add_text(text[half:])
        res = dm.finish()
    _cmp(res, nout, cnt, mats)
    # the sums are exact 128-bit fixed-point sums, rounded once (distmat.hip, Fix128): they do not depend on how the lines are cut
    # into batches, on the order the blocks run in, or on the run -- bit for bit
    with pydsm_mod.DistMat(smpls, maxent=maxent, minfreq=minfreq, run_to_sample=mapping, sizes=sizes) as dm:
        cuts = [0] + [text.rfind(b"\n", 0, len(text) * k // 5) + 1 for k in range(1, 5)] + [len(text)]
        for a, b in zip(cuts, cuts[1:]):
            if b > a:
                dm.add_text(text[a:b])
        res2 = dm.finish()
    for k in ("count", "log", "sqrt", "lgamma"):
        assert np.array_equal(res[k], res2[k]), k
    got = pydsm_mod.DistMat.format(res)
    assert got[0] == texts[0]             # the count file is exact
    for g, w in zip(got[1:], texts[1:]):  # the double files agree line for line up to the last printed digits
        assert g.count(b"\n") == w.count(b"\n")
        for a, b in zip(g.split(), w.split()):
            if a != b:
                assert abs(float(a) - float(b)) <= 1e-9 * max(1.0, abs(float(b))) + 2e-6, (a, b)


@pytest.mark.gpu
def test_gpu_distmat_fused_with_mining_and_many_samples(golden, pydsm_mod):
    """Tuples go from the miner's sink straight into the accumulator (no text round trip); 30 samples use the global-atomic
    path (the matrices do not fit LDS)."""
    for setname, prefixes, kw, maxent in (("five", ["A", "C", "G", "T"], dict(fmin=10, emax=2.0), [0.4, 0.8, 1.0]),
                                         ("many30", ["AC", "G"], dict(fmin=3, maxdepth=14, emax=5.0), [0.7, 0.9, 1.0])):
        names = golden.manifest["sets"][setname]["names"]
        idx = [pydsm_mod.Index(golden.fmi(setname, n)) for n in names]
        with pydsm_mod.DistMat(len(names), maxent=maxent) as dm, pydsm_mod.Miner(idx, **kw) as m:
            text, st = m.mine_many(prefixes, on_batch=dm.add)
            res = dm.finish()
        texts, nout, cnt, mats = oracle_run(text, len(names), maxent, 0)
        _cmp(res, nout, cnt, mats)
        assert int(res["noutput"][0]) <= st.tuples
        for ix in idx:
            ix.close()


@pytest.mark.gpu
def test_gpu_distmat_large_frequencies_and_errors(pydsm_mod):
    rng = np.random.default_rng(5)
    lines = []
    for _ in range(20000):
        k = int(rng.integers(1, 7))
        ids = rng.choice(6, k, replace=False)
        fr = [int(x) for x in np.where(rng.random(k) < 0.02, rng.integers(100000, 3000000, k), rng.integers(1, 400, k))]
        lines.append("ACGT 0.5 " + " ".join("%d:%d" % (i, f) for i, f in zip(ids, fr)))
    text = ("\n".join(lines) + "\n").encode()
    maxent = [0.2, 0.5, 0.9, 1.0]
    texts, nout, cnt, mats = oracle_run(text, 6, maxent, 3)
    with pydsm_mod.DistMat(6, maxent=maxent, minfreq=3) as dm:
        dm.add_text(text)
        _cmp(dm.finish(), nout, cnt, mats)
    with pytest.raises(pydsm_mod.DsmError):
        pydsm_mod.DistMat(1, maxent=[1.0])
    with pytest.raises(pydsm_mod.DsmError):
        pydsm_mod.DistMat(4, maxent=[1.5])
    with pydsm_mod.DistMat(3, maxent=[1.0]) as dm:
        with pytest.raises(pydsm_mod.DsmError):
            dm.add_text(b"ACG 0.1 7:3\n")          # sample id out of range


@pytest.mark.gpu
def test_cli_dropin_writes_the_tools_files(tmp_path):
    """smtxt2entropy_hip with the reference's options: the count file is byte-identical to the reference tool's, the double
    files agree to the printed precision; an existing output file is refused like the tool does."""
    exe = os.path.join(ROOT, "dsm-framework_amd", "host", "smtxt2entropy_hip")
    text, smpls, maxent, minfreq, _, _ = case_input("five", "minfreq")
    args = [exe, "-s", str(smpls), "-m", ",".join(str(x) for x in maxent), "-M", str(minfreq), "-F", "o"]
    r = subprocess.run(args, input=text, cwd=tmp_path, capture_output=True)
    assert r.returncode == 0, r.stderr
    for kind in ("count", "log", "sqrt", "lgamma"):
        got = open(os.path.join(tmp_path, kind + ".o"), "rb").read()
        want = gzip.open(os.path.join(GOLD, "five", "distmat.minfreq.%s.gz" % kind), "rb").read()
        if kind == "count":
            assert got == want
        else:
            assert len(got.split()) == len(want.split())
            for a, b in zip(got.split(), want.split()):
                if a != b:
                    assert abs(float(a) - float(b)) <= 1e-9 * max(1.0, abs(float(b))) + 2e-6, (kind, a, b)
    r = subprocess.run(args, input=text, cwd=tmp_path, capture_output=True)
    assert r.returncode == 1 and b"already exists" in r.stderr
    # -e steps
    text, smpls, maxent, _, _, _ = case_input("toy3", "step")
    r = subprocess.run([exe, "-s", str(smpls), "-e", "0.3", "-F", "s"], input=text, cwd=tmp_path, capture_output=True)
    assert r.returncode == 0, r.stderr
    assert open(os.path.join(tmp_path, "count.s"), "rb").read() == gzip.open(os.path.join(GOLD, "toy3", "distmat.step.count.gz"), "rb").read()


@pytest.mark.gpu
def test_cli_samplefile_and_normalize(tmp_path):
    """-S and -N through the drop-in CLI, files in the tool's formats."""
    exe = os.path.join(ROOT, "dsm-framework_amd", "host", "smtxt2entropy_hip")
    text, smpls, maxent, minfreq, mapping, sizes = case_input("five", "smap_norm")
    open(os.path.join(tmp_path, "map.txt"), "w").write("".join("%d\n" % x for x in mapping))
    open(os.path.join(tmp_path, "sizes.txt"), "w").write("".join("d%d\t%r\n" % (i, x) for i, x in enumerate(sizes)))
    r = subprocess.run([exe, "-S", "map.txt", "-N", "sizes.txt", "-e", "0.5", "-F", "o"], input=text, cwd=tmp_path, capture_output=True)
    assert r.returncode == 0, r.stderr
    for kind in ("count", "log", "sqrt", "lgamma"):
        got = open(os.path.join(tmp_path, kind + ".o"), "rb").read()
        want = gzip.open(os.path.join(GOLD, "five", "distmat.smap_norm.%s.gz" % kind), "rb").read()
        if kind in ("count", "lgamma"):      # lgamma stays zero under -N
            assert got == want, kind
        else:
            for a, b in zip(got.split(), want.split()):
                if a != b:
                    assert abs(float(a) - float(b)) <= 1e-9 * max(1.0, abs(float(b))) + 2e-6, (kind, a, b)
    r = subprocess.run([exe, "-s", "3", "-S", "map.txt", "-m", "1.0", "-F", "x"], input=text, cwd=tmp_path, capture_output=True)
    assert r.returncode == 1
