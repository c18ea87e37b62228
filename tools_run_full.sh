set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
$R/tools/gather_calib 4096 268435456 64 | tee $R/gpurun_out/calib_plain.txt
$R/tools/gather_calib 4096 134217728 128 | tee -a $R/gpurun_out/calib_plain.txt
rocprofv3 -L 2>/dev/null | grep -i -E "FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ|TCC_HIT|TCC_MISS|TCC_REQ" | head -20 > $R/gpurun_out/counters.txt || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_calib64 -- $R/tools/gather_calib 4096 268435456 64 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_calib128 -- $R/tools/gather_calib 4096 134217728 128 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_pmc_write.log 2>&1
ls $R/gpurun_out/pmc_fetch/*/ | head
