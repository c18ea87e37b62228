# Per-kernel stats of the bench command, with the SAME --steps / --warmup as the quoted bench line (default: 3 / 1), and the
# LF-step kernel summed over ALL its instantiations per pass, averaged over the real launches only (the launches queued ahead
# that found another frequency class return at once).  kernel_stats.sh [STEPS] [WARMUP]
set -e
R=$GRAFT_REPO_ROOT
STEPS=${1:-3}; WARM=${2:-1}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu --no-extras > $R/gpurun_out/bench_prof.log 2>&1
f=$(find $R/gpurun_out/prof_cur -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/kernel_stats_cur.csv
python3 $R/tools/profiling/make_traffic.py trace $(find $R/gpurun_out/prof_cur -name "*kernel_trace.csv" | head -1) $STEPS $WARM
find $R/gpurun_out/prof_cur -name "*.csv" -size +1M -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
passes=$STEPS+$WARM
for r in rows[:24]:
    print("%-60s %6s %10.3f ms/pass %8.1f us" % (r["Name"][:60], r["Calls"], int(r["TotalDurationNs"])/1e6/passes, float(r["AverageNs"])/1e3))
PY
grep '^{"metric' $R/gpurun_out/bench_prof.log | cut -c1-400
