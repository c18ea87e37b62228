import sys, time, glob, os
sys.path.insert(0, "dsm-framework_amd")
import torch, pydsm
from pydsm import builder
d = "/tmp/dsm_bench"; os.makedirs(d, exist_ok=True)
paths = []
for s in range(4):
    p = os.path.join(d, "srv-%d.fmi" % s)
    if not os.path.exists(p):
        codes = builder.synth_reads(200 + s, 1000000, 100, 5000000, 0.005, device="cuda", private_frac=0.05)
        builder.build_from_codes(codes, p)
    paths.append(p)
idx = [pydsm.Index(p) for p in paths]
streams = []
t0 = time.time()
for ix in idx:
    with pydsm.Miner([ix], fmin=10, stream_mode=True) as m:
        b, st = m.enumerate("A", with_header=False)
    streams.append(b)
print("client streams: %.1f MB each, %.2f s total" % (len(streams[0]) / 1e6, time.time() - t0), flush=True)
t0 = time.time()
tries = [pydsm.Trie(b) for b in streams]
t1 = time.time()
print("parse: %.2f s = %.1f MB/s per stream" % (t1 - t0, sum(len(b) for b in streams) / 1e6 / (t1 - t0)), flush=True)
txt, st = pydsm.merge(tries, pmin=2, emax=2.0, text=False)
t2 = time.time()
print("merge: %.2f s, union nodes %d, tuples %d" % (t2 - t1, st.union_nodes, st.tuples))
t0 = time.time()
_, st2 = pydsm.mine(idx, "A", fmin=10, pmin=2, emax=2.0, text=False)
print("fused mine of the same prefix: %.2f s, tuples %d" % (time.time() - t0, st2.tuples))
