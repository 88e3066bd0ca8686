cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r2h/tests.log 2>&1; echo "rc $?" >> gpurun_out/r2h/tests.log; tail -30 gpurun_out/r2h/tests.log
