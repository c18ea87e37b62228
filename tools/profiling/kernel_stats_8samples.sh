set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --nlocal 8  > $R/gpurun_out/bench8_plain.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_8 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --nlocal 8  > $R/gpurun_out/bench_prof8.log 2>&1
f=$(find $R/gpurun_out/prof_8 -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/kernel_stats_8.csv
find $R/gpurun_out/prof_8 -name "*.csv" -size +1M -delete
tail -1 $R/gpurun_out/bench_prof8.log | cut -c1-300
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/kernel_stats_8.csv")))
for r in rows:
    if r["Name"].startswith("dsm::") or "dsm::" in r["Name"][:12]:
        print("%-60s %6s %10.3f ms %8.1f us" % (r["Name"][:60], r["Calls"], int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
