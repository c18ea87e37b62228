set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --nlocal 8 --reads 2000000 --genome 10000000 > $R/gpurun_out/bench8_plain.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_8 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --nlocal 8 --reads 2000000 --genome 10000000 > $R/gpurun_out/bench_prof8.log 2>&1
f=$(find $R/gpurun_out/prof_8 -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/kernel_stats_8.csv
find $R/gpurun_out/prof_8 -name "*.csv" -size +1M -delete
tail -1 $R/gpurun_out/bench_prof8.log | cut -c1-300
