set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_gaps -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/bench_gaps.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "dsm::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last third (timed steps), main-stream kernels only (skip copies on the copy stream: names with copyBuffer)
rows = [r for r in rows if "copyBuffer" not in r["Kernel_Name"]]
n = len(rows)
rows = rows[n // 3:]
busy = 0; gaps = 0; biggaps = 0; last_end = None; after = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s > last_end:
        g = s - last_end
        if g < 5_000_000:
            gaps += g
            key = prev.split("(")[0][-40:]
            after[key] = after.get(key, 0) + g
    busy += e - s
    last_end = max(e, last_end or 0)
    prev = r["Kernel_Name"]
print("kernels", len(rows), "busy ms", busy / 1e6, "gaps ms", gaps / 1e6)
for k, v in sorted(after.items(), key=lambda kv: -kv[1])[:12]:
    print("  gap after %-42s %.2f ms" % (k, v / 1e6))
PY
rm -rf $R/gpurun_out/prof_gaps
