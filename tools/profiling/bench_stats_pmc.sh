set -e
R=$GRAFT_REPO_ROOT
python $R/bench.py 2> $R/gpurun_out/bench_final.err | tee $R/gpurun_out/bench_final.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01j -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/bench_prof.log 2>&1
cp $(find $R/gpurun_out/prof_r01j -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats_v8.csv
echo stats done
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc8_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_pmc_$n.log 2>&1 || echo "fail $n"
  python3 $R/tools/profiling/pmc_agg.py $R/gpurun_out/pmc8_$n >> $R/gpurun_out/pmc8_summary.txt
  echo "pmc $n done"
done
find $R/gpurun_out/prof_r01j $R/gpurun_out/pmc8_* -name "*.csv" -size +1M -delete
cat $R/gpurun_out/pmc8_summary.txt
