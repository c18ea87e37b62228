set -e
python bench.py --steps 1 --warmup 0 --cpu-seconds 15 > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err
tail -c 2500 gpurun_out/bench_full.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_r01 | head -20
