# SQ instruction / wait counters of the default bench command (two --pmc passes, no trace domains combined with --pmc)
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-cur}
cd /tmp && export TMPDIR=/tmp
i=0
rm -f $R/gpurun_out/pmcsq_${TAG}_summary.txt
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq_${TAG}_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/pmcsq_${TAG}_$i.log 2>&1 || echo "fail $c"
  python3 $R/tools/profiling/pmc_agg.py $R/gpurun_out/pmcsq_${TAG}_$i >> $R/gpurun_out/pmcsq_${TAG}_summary.txt 2>&1 || true
  rm -rf $R/gpurun_out/pmcsq_${TAG}_$i
  echo "pass $i done"
done
