cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "config5" --durations=5 > gpurun_out/r2f/c5.log 2>&1; tail -15 gpurun_out/r2f/c5.log
