"""Cost of creating a miner (device buffers are allocated once per miner) against the cost of mining one prefix, for
several arena budgets.  Uses the four 1M-read samples of server_side_bench.py."""
import sys, time, os
sys.path.insert(0, "dsm-framework_amd")
import torch, pydsm
paths = ["/tmp/dsm_bench/srv-%d.fmi" % s for s in range(4)]
idx = [pydsm.Index(p) for p in paths]
for budget in (0, 100 << 30, 40 << 30, 10 << 30, 4 << 30):
    t0 = time.time()
    m = pydsm.Miner(idx, fmin=10, pmin=2, emax=2.0, arena_bytes=budget)
    t1 = time.time()
    _, st = m.mine("A", text=False)
    t2 = time.time()
    _, st = m.mine("A", text=False)
    t3 = time.time()
    m.close()
    print("arena_bytes %5.0f GiB: create %.2f s, first mine %.3f s, second mine %.3f s, tuples %d splits %d" % (budget / 2**30, t1 - t0, t2 - t1, t3 - t2, st.tuples, st.splits), flush=True)
