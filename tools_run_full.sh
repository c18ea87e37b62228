set -e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
tail -c 400 $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log
