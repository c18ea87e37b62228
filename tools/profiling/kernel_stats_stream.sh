set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/prof_stream -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --stream-mode > $R/gpurun_out/bench_prof_stream.log 2>&1
f=$(find $R/gpurun_out/prof_stream -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/kernel_stats_stream.csv
python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open("$f")))
for r in rows[:16]:
    print("%-60s %6s %10.3f ms %8.1f us" % (r["Name"][:60], r["Calls"], int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
for g in glob.glob("$R/gpurun_out/prof_stream/**/*memory_copy_stats.csv", recursive=True):
    for r in csv.DictReader(open(g)): print(r)
PY
find $R/gpurun_out/prof_stream -name "*.csv" -size +1M -delete
