"""BASELINE's "minlen=10" has no literal counterpart in the reference (the only 10 is --fmin); the server's
-m/--mindepth is a pure output filter.  Reports the tuple counts of the benchmark pass with and without -m 10."""
import glob
import sys
sys.path.insert(0, "dsm-framework_amd")
import pydsm
p = sorted(glob.glob("/tmp/dsm_bench/sample-0.*r10000000*.fmi"))[0]
ix = pydsm.Index(p)
for md in (0, 10):
    with pydsm.Miner([ix], fmin=10, pmin=1, emax=2.0, mindepth=md) as m:
        _, st = m.mine_many(["A", "C", "G", "T"], text=False)
    print("mindepth %2d: nodes %d tuples %d candidates %d" % (md, st.reported, st.tuples, st.candidates))
ix.close()
