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

# ---- merge while receiving (dsm_server_*): four reader threads feeding 1 MB pieces, device memory sampled on the way ----
import threading, hashlib
def run_server(prefix_len, unit_extra):
    h = hashlib.sha256()
    ntup = [0]
    def on_batch(b):
        ntup[0] += int(b.ntuples)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    low = [free0]
    stop = [False]
    def sampler():
        while not stop[0]:
            low[0] = min(low[0], torch.cuda.mem_get_info()[0])
            time.sleep(0.002)
    srv = pydsm.Server(len(streams), prefix_len=prefix_len, unit_extra=unit_extra, pmin=2, emax=2.0, text=False, on_batch=on_batch)
    ts = threading.Thread(target=sampler); ts.start()
    t0 = time.time()
    def reader(i):
        b = streams[i]
        for o in range(0, len(b), 1 << 20):
            srv.feed(i, b[o:o + (1 << 20)])
        srv.end(i)
    ths = [threading.Thread(target=reader, args=(i,)) for i in range(len(streams))]
    for t in ths: t.start()
    for t in ths: t.join()
    t1 = time.time()
    _, st = srv.finish()
    t2 = time.time()
    stop[0] = True; ts.join()
    units, peak = srv.units()
    srv.close()
    return dict(feed_s=t1 - t0, finish_s=t2 - t1, tuples=st.tuples, units=units, peak_unit_nodes=peak, device_peak_mb=(free0 - low[0]) / 1e6)
for tr in tries:
    tr.close()
for plen, extra in ((None, 0), (1, 0), (1, 1), (1, 2)):
    r = run_server(plen, extra)
    print("server prefix_len=%s unit_extra=%d: feed %.2f s, finish %.2f s, tuples %d, units %d (largest %d nodes), device memory in use at most %.0f MB" % (
        plen, extra, r["feed_s"], r["finish_s"], r["tuples"], r["units"], r["peak_unit_nodes"], r["device_peak_mb"]), flush=True)
