# Instruction counts by class and wave residency of the LF-step kernel (profiles/r03_lf_bound.txt, sections 1-2): one rocprofv3 --pmc
# pass of the default bench command.  usage: bash tools/profiling/pmc_insts.sh [tag]   (DSM_LIB_PATH picks a library variant)
R=$GRAFT_REPO_ROOT
TAG=${1:-new}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > /dev/null 2>&1
rm -f $R/gpurun_out/pmc3_${TAG}_summary.txt
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc3_${TAG}_$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $R/gpurun_out/pmc3_${TAG}_$i.log 2>&1 || echo "fail $c"
  python3 $R/tools/profiling/pmc_agg.py $R/gpurun_out/pmc3_${TAG}_$i >> $R/gpurun_out/pmc3_${TAG}_summary.txt 2>&1 || true
  rm -rf $R/gpurun_out/pmc3_${TAG}_$i
done
python3 - <<PY
import json
for line in open('$R/gpurun_out/pmc3_${TAG}_summary.txt'):
    d=json.loads(line)
    for k,v in sorted(d.items()):
        if 'expand_kernel' in k and v['n']>300: print('$TAG', k[-50:], '%.4g'%v['sum'], v['n'])
PY
