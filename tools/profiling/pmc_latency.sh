# Memory-side counters of the LF-step kernel (profiles/r03_lf_bound.txt, section 1): TCP->TCC read / write latency, vector-memory
# instruction counts, TA FIFO stalls, one rocprofv3 --pmc pass per group of the default bench command (no trace domains combined with
# --pmc).  The UTCL1 and TA_BUSY groups are refused by this rocprofv3 build ("fail ..." is printed and the pass skipped).
# usage: bash tools/profiling/pmc_latency.sh [tag]   (DSM_LIB_PATH picks a library variant)
R=$GRAFT_REPO_ROOT
TAG=${1:-base}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > /dev/null 2>&1
i=0
rm -f $R/gpurun_out/pmc2_${TAG}_summary.txt
for c in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_${TAG}_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $R/gpurun_out/pmc2_${TAG}_$i.log 2>&1 || echo "fail $c"
  python3 $R/tools/profiling/pmc_agg.py $R/gpurun_out/pmc2_${TAG}_$i >> $R/gpurun_out/pmc2_${TAG}_summary.txt 2>&1 || true
  rm -rf $R/gpurun_out/pmc2_${TAG}_$i
  echo "pass $i done"
done
