R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in wt_2e11ea0 wt_3979313; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -- python3 $R/scratch/$w/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_$w.log 2>&1
  f=$(find $R/gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/kernel_stats_$w.csv
  rm -rf $R/gpurun_out/prof_$w
  echo $w; grep "advance_down\|cand_store\|expand_kernel\|fillBuffer\|filter_kernel" $f | cut -d, -f1-4 | cut -c1-40,150-
done
