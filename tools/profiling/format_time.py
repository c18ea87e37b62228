"""Cost of the reference's text formatting (dsm_format_batch: "path %f id:freq ...\n", metaserver.cpp:472-484) next to
mining the same prefix of the benchmark index with a binary sink."""
import glob, sys, time
sys.path.insert(0, "dsm-framework_amd")
import pydsm
p = sorted(glob.glob("/tmp/dsm_bench/sample-0.*r10000000*e0.005.fmi"))[0]
ix = pydsm.Index(p)
with pydsm.Miner([ix], fmin=10, pmin=1, emax=2.0) as m:
    m.mine("C", text=False)
    t0 = time.time(); _, st = m.mine("A", text=False); t1 = time.time()
    txt, st2 = m.mine("A", text=True); t2 = time.time()
print("binary sink: %.3f s; with text formatting: %.3f s for %d tuples, %.1f MB of text = %.1f M lines/s" %
      (t1 - t0, t2 - t1, st2.tuples, len(txt) / 1e6, st2.tuples / (t2 - t1) / 1e6))
