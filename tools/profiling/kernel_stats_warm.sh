set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/bench_prof_w.log 2>&1
cp $(find $R/gpurun_out/prof_r01i -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats_v7w.csv
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_r01i/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "expand_kernel" in r["Kernel_Name"]]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
n=len(d)//4
print("expand launches", len(d), "all avg us", sum(d)/len(d)/1e3, "warmup pass avg", sum(d[:n])/n/1e3, "timed passes avg", sum(d[n:])/(len(d)-n)/1e3)
PY
find $R/gpurun_out/prof_r01i -name "*.csv" -size +1M -delete
grep '^{"metric' $R/gpurun_out/bench_prof_w.log | cut -c1-900
