set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $R/gpurun_out/t.log 2>&1 || true
tail -3 $R/gpurun_out/t.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --nlocal 8 --reads 2000000 --genome 10000000 --prefix-len 1 2> $R/gpurun_out/b8.err | tee $R/gpurun_out/b8.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d8b -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --nlocal 8 --reads 2000000 --genome 10000000 --prefix-len 1 > $R/gpurun_out/bench_prof8.log 2>&1
