"""A/B timing of library variants on the configs[1] workload (one sample of 10^7 x 100 bp, four one-letter prefixes): a hash of
prefix G's tuples for parity, then wall / device / LF-step times per pass.
usage: python tools/profiling/ab.py [reps] ; DSM_LIB_PATH picks the variant (pydsm loads it instead of the in-tree library)."""
import hashlib, os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, pydsm
from pydsm import builder
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reads = int(os.environ.get("AB_READS", "10000000"))
d = "/tmp/dsm_bench"; os.makedirs(d, exist_ok=True)
path = os.path.join(d, "sample-0.s42_r%d_l100_g%d_e0.005.fmi" % (reads, reads * 5))
if not os.path.exists(path):
    codes = builder.synth_reads(42, reads, 100, reads * 5, 0.005, device="cuda")
    builder.build_from_codes(codes, path + ".tmp"); del codes; torch.cuda.empty_cache(); os.replace(path + ".tmp", path)
ix = pydsm.Index(path, device=0)
hp, hl, hf = hashlib.sha256(), hashlib.sha256(), hashlib.sha256()   # paths, lengths, frequencies: each over the whole prefix, so that
cnt = [0]                                                            # the hash does not depend on where the batches are cut
def on_batch(b):
    n = int(b.ntuples); cnt[0] += n
    po = np.ctypeslib.as_array(b.path_off, shape=(n + 1,)); qo = np.ctypeslib.as_array(b.pair_off, shape=(n + 1,))
    hp.update(np.ctypeslib.as_array(ctypes.cast(b.path_bytes, ctypes.POINTER(ctypes.c_uint8)), shape=(int(po[-1]),)).tobytes())
    hl.update(np.diff(po).astype(np.uint32).tobytes())
    hf.update(np.ctypeslib.as_array(b.freqs, shape=(int(qo[-1]),)).tobytes())
with pydsm.Miner([ix], fmin=10, pmin=1, emax=2.0) as m:
    _, st = m.mine_many(["G"], text=False, on_batch=on_batch)
    print("parity G: tuples %d nodes %d sha %s" % (cnt[0], st.reported, hashlib.sha256(hp.digest() + hl.digest() + hf.digest()).hexdigest()[:16]))
    m.mine_many(["A", "C", "G", "T"], text=False)
    best = None
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = m.mine_many(["A", "C", "G", "T"], text=False)[1]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        line = "wall %.1f ms  device %.1f  expand %.1f (%.1f us/launch, %d)  non-LF %.1f  host %.1f  nodes %.4e tuples %d" % (
            dt, st.device_ms, st.expand_ms, st.expand_ms * 1e3 / max(1, st.expand_launches), st.expand_launches, st.device_ms - st.expand_ms, st.host_ms, st.reported, st.tuples)
        print(os.environ.get("AB_TAG", "lib"), line, flush=True)
ix.close()
