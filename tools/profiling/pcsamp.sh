# PC sampling of the default bench command (beta feature of rocprofv3): where do the waves of the LF-step kernel spend their time?
R=$GRAFT_REPO_ROOT
METHOD=${1:-stochastic}
UNIT=${2:-cycles}
INTERVAL=${3:-1048576}
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL --kernel-trace --output-format csv -d $R/gpurun_out/pcs -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/pcs.log 2>&1
echo rc=$?
tail -5 $R/gpurun_out/pcs.log
find $R/gpurun_out/pcs -type f | head; du -sh $R/gpurun_out/pcs
