# Round-3 evidence, all from ONE box: the bench line, the kernel trace of the SAME command (--steps 3 --warmup 1), and the HBM
# traffic of the LF-step kernel (rocprofv3 --pmc, one pass per counter group, no trace domains combined with --pmc).
# Writes gpurun_out/r03_*; tools/profiling/make_traffic.py turns them into profiles/traffic.json + profiles/r03_kernel_stats.csv.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
STEPS=3; WARM=1
python3 $R/bench.py --no-extras --steps $STEPS --warmup $WARM > $R/gpurun_out/r03_bench.json 2> $R/gpurun_out/r03_bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu --no-extras > $R/gpurun_out/r03_prof.log 2>&1
cp $(find $R/gpurun_out/r03_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r03_kernel_stats.csv
python3 $R/tools/profiling/make_traffic.py trace $(find $R/gpurun_out/r03_prof -name "*kernel_trace.csv" | head -1) $STEPS $WARM > $R/gpurun_out/r03_trace_summary.json
find $R/gpurun_out/r03_prof -name "*.csv" -size +1M -delete
echo "stats done"
rm -f $R/gpurun_out/r03_pmc.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $R/gpurun_out/r03_pmc_$n.log 2>&1 || echo "fail $n"
  python3 $R/tools/profiling/pmc_agg.py $R/gpurun_out/r03_pmc_$n >> $R/gpurun_out/r03_pmc.txt
  rm -rf $R/gpurun_out/r03_pmc_$n
  echo "pmc $n done"
done
python3 $R/tools/profiling/make_traffic.py traffic $R/gpurun_out/r03_pmc.txt $R/gpurun_out/r03_bench.json $R/gpurun_out/r03_trace_summary.json > $R/gpurun_out/r03_traffic.json
cat $R/gpurun_out/r03_traffic.json
