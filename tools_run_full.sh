set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q > $R/gpurun_out/t_full.log 2>&1 || true
tail -15 $R/gpurun_out/t_full.log
