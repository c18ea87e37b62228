set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_prof.log 2>&1
f=$(find $R/gpurun_out/prof_cur -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/kernel_stats_cur.csv
find $R/gpurun_out/prof_cur -name "*.csv" -size +1M -delete
cut -c1-60 $f | head -5
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:24]:
    print("%-60s %6s %10.3f ms %8.1f us" % (r["Name"][:60], r["Calls"], int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
