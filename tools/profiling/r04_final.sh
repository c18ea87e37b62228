# Round-4 evidence, all from ONE box: the bench line, the kernel trace of the SAME command (--steps 3 --warmup 1), and the HBM
# traffic of the LF-step kernel (rocprofv3 --pmc with --kernel-trace only -- never combined with the sys / hip / hsa / memory-copy
# trace domains -- one pass per counter group).
# Writes gpurun_out/r04_*; tools/profiling/make_traffic.py turns them into profiles/traffic.json.
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun: GRAFT_REPO_ROOT is the root of the snapshot}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out"
mkdir -p "$O"
STEPS=3; WARM=1
python3 "$R/bench.py" --no-extras --steps $STEPS --warmup $WARM > "$O/r04_bench.json" 2> "$O/r04_bench.err"
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/r04_prof" -- python3 "$R/bench.py" --steps $STEPS --warmup $WARM --no-cpu --no-extras > "$O/r04_prof.log" 2>&1
cp "$(find "$O/r04_prof" -name "*kernel_stats.csv" | head -1)" "$O/r04_kernel_stats.csv"
REAL=$(python3 -c "import json,sys; j=json.loads([l for l in open('$O/r04_bench.json') if l.startswith('{')][-1]); print(j['roofline']['launches'] // j['steps'])")
python3 "$R/tools/profiling/make_traffic.py" trace "$(find "$O/r04_prof" -name "*kernel_trace.csv" | head -1)" $STEPS $WARM 0 $REAL > "$O/r04_trace_summary.json"
find "$O/r04_prof" -name "*.csv" -size +1M -delete
echo "stats done"
rm -f "$O/r04_pmc.txt"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  n=$(echo "$c" | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/r04_pmc_$n" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --no-extras > "$O/r04_pmc_$n.log" 2>&1 || echo "fail $n"
  python3 "$R/tools/profiling/pmc_agg.py" "$O/r04_pmc_$n" >> "$O/r04_pmc.txt"
  rm -rf "$O/r04_pmc_$n"
  echo "pmc $n done"
done
python3 "$R/tools/profiling/make_traffic.py" traffic "$O/r04_pmc.txt" "$O/r04_bench.json" "$O/r04_trace_summary.json" > "$O/r04_traffic.json"
cat "$O/r04_traffic.json"
# ---- eight samples on the one card (dense LF-step sweeps): kernel stats of one record's two passes, counters of the dense kernel ----
cd "$R" && python3 "$R/bench.py" --steps 1 --warmup 0 --cpu-seconds 1 --extras d8 > "$O/r04_d8_bench.json" 2> /dev/null   # (builds the indexes; the line itself is kept)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/r04_prof_d8" -- python3 "$R/bench.py" --steps 1 --warmup 0 --cpu-seconds 1 --extras d8 > "$O/r04_prof_d8.log" 2>&1
cp "$(find "$O/r04_prof_d8" -name "*kernel_stats.csv" | head -1)" "$O/r04_kernel_stats_8samples.csv"
rm -rf "$O/r04_prof_d8"
rm -f "$O/r04_pmc_d8.txt"
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$(echo "$c" | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/r04_pmc8_$n" -- python3 "$R/bench.py" --steps 1 --warmup 0 --cpu-seconds 1 --extras d8 > "$O/r04_pmc8_$n.log" 2>&1 || echo "fail $n"
  python3 "$R/tools/profiling/pmc_agg.py" "$O/r04_pmc8_$n" >> "$O/r04_pmc_d8.txt"
  rm -rf "$O/r04_pmc8_$n"
  echo "pmc d8 $n done"
done
