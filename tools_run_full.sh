set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_l2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_pmc_l2.log 2>&1 || true
echo done
