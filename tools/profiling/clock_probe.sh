# Builds the library with -DDSM_CLOCK_PROBE (the LF-step kernel then records, per launch, the shader clock against the 100 MHz
# real-time counter, the earliest / latest wave start and end and -- for the 51st launch of a prefix -- every wave's end with its
# XCC / SE / CU / SIMD ids) and runs the A/B harness with it.  The probe's atomics cost ~15 % of the kernel's time: the figures it
# prints are ratios, not timings.  usage (on the GPU box): bash tools/profiling/clock_probe.sh
set -e
R=$GRAFT_REPO_ROOT
cd $R/dsm-framework_amd
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DDSM_CLOCK_PROBE -c csrc/engine.hip -o /tmp/engine_probe.o
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdsmhip_probe.so csrc/index.o /tmp/engine_probe.o csrc/distmat.o csrc/bwt.o csrc/exchange.o csrc/fmiwrite.o -lpthread -ldl
cd $R
DSM_LIB_PATH=/tmp/libdsmhip_probe.so AB_TAG=probe python tools/profiling/ab.py 1 2>&1 | grep -E "probe|clock" | cut -c1-1200
