R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_64 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --nlocal 64 --reads 100000 --genome 500000 --pmin 1 --pmax 1 > $R/gpurun_out/bench_prof64.log 2>&1
f=$(find $R/gpurun_out/prof_64 -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/kernel_stats_64.csv
rm -rf $R/gpurun_out/prof_64
grep "^{" $R/gpurun_out/bench_prof64.log | cut -c1-200
