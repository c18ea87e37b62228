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
    return text, len(MAN["sets"][setname]["names"]), maxent, minfreq


def oracle_run(text, smpls, maxent, minfreq):
    L = _olib()
    me = (C.c_double * len(maxent))(*maxent)
    h = L.orc_distmat_new(smpls, me, len(maxent), minfreq)
    assert h
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
    text, smpls, maxent, minfreq = case_input(setname, case)
    texts, nout, cnt, mats = oracle_run(text, smpls, maxent, minfreq)
    for kind, got in zip(("count", "log", "sqrt", "lgamma"), texts):
        want = gzip.open(os.path.join(GOLD, setname, "distmat.%s.%s.gz" % (case, kind)), "rb").read()
        assert got == want, (setname, case, kind)
    assert nout.max() <= text.count(b"\n")
