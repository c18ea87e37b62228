# one rocprofv3 --pmc pass of the default bench command: pmc_one.sh TAG "COUNTER COUNTER ..."
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $@ --kernel-trace --output-format csv -d $R/gpurun_out/pmc1_${TAG} -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/pmc1_${TAG}.log 2>&1 || echo "fail"
python3 $R/tools/profiling/pmc_agg.py $R/gpurun_out/pmc1_${TAG} > $R/gpurun_out/pmc1_${TAG}.json
rm -rf $R/gpurun_out/pmc1_${TAG}
python3 - <<PY
import json
d=json.loads(open("$R/gpurun_out/pmc1_${TAG}.json").read().splitlines()[0])
for k,v in d.items():
    if "expand_kernel" in k: print(k[-45:], "%.4g"%v["sum"])
PY
