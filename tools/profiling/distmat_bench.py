#!/usr/bin/env python3
"""Throughput of the distance-matrix accumulation on synthetic tuple lines: the unmodified reference tool
(oracle/_ref/smtxt2entropy, one CPU thread -- it has no parallel mode) against smtxt2entropy_hip on the same text, and the
in-process API fed with binary batches.  usage: distmat_bench.py [lines=3000000] [samples=8]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rng = np.random.default_rng(1)
    k = rng.integers(1, s + 1, n)
    lines = []
    for i in range(n):
        ids = rng.choice(s, int(k[i]), replace=False)
        fr = rng.integers(10, 60, int(k[i]))
        lines.append("ACGTACGTACGTACGTACGT 1.234567 " + " ".join("%d:%d" % (a, b) for a, b in zip(ids, fr)))
    text = ("\n".join(lines) + "\n").encode()
    print("input: %d lines, %d samples, %.1f MB" % (n, s, len(text) / 1e6), flush=True)
    with tempfile.TemporaryDirectory() as td:
        ref = os.path.join(ROOT, "oracle", "_ref", "smtxt2entropy")
        if os.path.exists(ref):
            t0 = time.time()
            subprocess.run([ref, "-s", str(s), "-m", "0.5,0.9,1.0", "-F", "r"], input=text, cwd=td, check=True, stderr=subprocess.DEVNULL)
            t = time.time() - t0
            print("reference tool (1 thread): %.2f s = %.2f M lines/s" % (t, n / t / 1e6), flush=True)
        exe = os.path.join(ROOT, "dsm-framework_amd", "host", "smtxt2entropy_hip")
        t0 = time.time()
        subprocess.run([exe, "-s", str(s), "-m", "0.5,0.9,1.0", "-F", "g"], input=text, cwd=td, check=True)
        t = time.time() - t0
        print("smtxt2entropy_hip (text in, whole process incl. GPU start-up): %.2f s = %.2f M lines/s" % (t, n / t / 1e6), flush=True)
        if os.path.exists(ref):
            a = open(os.path.join(td, "count.r"), "rb").read()
            b = open(os.path.join(td, "count.g"), "rb").read()
            print("count files identical:", a == b)
    import pydsm
    dm = pydsm.DistMat(s, maxent=[0.5, 0.9, 1.0])
    dm.add_text(text[: text.find(b"\n", 1 << 16) + 1])  # warm up
    t0 = time.time()
    dm.add_text(text)
    t = time.time() - t0
    print("dsm_distmat_add_text in process (parse + bucket on the host, terms on the GPU): %.2f s = %.2f M lines/s" % (t, n / t / 1e6))
    dm.close()
    # binary batches (what a tuple sink receives): no text parsing
    import ctypes as C
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum(k)
    # ids must be unique inside a tuple: rebuild them as a rotation per tuple
    base = np.repeat(rng.integers(0, s, n), k)
    within = np.arange(off[-1]) - np.repeat(off[:-1], k)
    ids = ((base + within) % s).astype(np.uint32)
    fr = rng.integers(10, 60, int(off[-1])).astype(np.uint64)
    b = pydsm.TupleBatch()
    b.ntuples = n
    b.pair_off = off.ctypes.data_as(C.POINTER(C.c_uint32))
    b.ids = ids.ctypes.data_as(C.POINTER(C.c_uint32))
    b.freqs = fr.ctypes.data_as(C.POINTER(C.c_uint64))
    dm = pydsm.DistMat(s, maxent=[0.5, 0.9, 1.0])
    dm.add(b)
    t0 = time.time()
    dm.add(b)
    t = time.time() - t0
    print("dsm_distmat_add (binary batch): %.3f s = %.2f M tuples/s" % (t, n / t / 1e6))
    dm.close()


if __name__ == "__main__":
    main()
