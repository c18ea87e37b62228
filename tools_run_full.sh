set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*Counter_Name|Name\s*:" | sed 's/.*:\s*//' | sort -u | tr '\n' ' ' > $R/gpurun_out/all_counters.txt || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc4_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_pmc_sq.log 2>&1 || echo sqfail
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc4_tc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/bench_pmc_tc.log 2>&1 || echo tcfail
echo done
