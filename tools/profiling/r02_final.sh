# Round-2 evidence: bench line, kernel trace stats and HBM traffic (PMC) of the same command.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python3 $R/bench.py --no-extras > $R/gpurun_out/r02_bench.json 2> $R/gpurun_out/r02_bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/r02_prof.log 2>&1
cp $(find $R/gpurun_out/r02_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02_kernel_stats.csv
find $R/gpurun_out/r02_prof -name "*.csv" -size +1M -delete
echo "stats done"
rm -f $R/gpurun_out/r02_pmc.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r02_pmc_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/r02_pmc_$n.log 2>&1 || echo "fail $n"
  python3 $R/tools/profiling/pmc_agg.py $R/gpurun_out/r02_pmc_$n >> $R/gpurun_out/r02_pmc.txt
  rm -rf $R/gpurun_out/r02_pmc_$n
  echo "pmc $n done"
done
