import sys, time, os, glob
sys.path.insert(0, "dsm-framework_amd")
import torch, pydsm
p = sorted(glob.glob("/tmp/dsm_bench/sample-0.*r10000000*.fmi"))
if not p:
    print("no cached index"); sys.exit(0)
p = p[0]
print(p, os.path.getsize(p) / 1e9, "GB")
for i in range(3):
    t0 = time.time(); ix = pydsm.Index(p); torch.cuda.synchronize(); dt = time.time() - t0
    print("open %d: %.3f s" % (i, dt)); ix.close()
