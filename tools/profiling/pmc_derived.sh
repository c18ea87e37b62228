set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for c in MeanOccupancyPerActiveCU VALUBusy "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcy_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/pmcy_$i.log 2>&1 || echo "fail $c"
  python3 $R/tools/profiling/pmc_sum.py $R/gpurun_out/pmcy_$i >> $R/gpurun_out/pmcy_summary.txt 2>&1 || true
  rm -rf $R/gpurun_out/pmcy_$i
done
cat $R/gpurun_out/pmcy_summary.txt
