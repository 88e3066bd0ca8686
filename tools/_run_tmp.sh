cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2l/tests.log 2>&1; echo "rc $?" >> gpurun_out/r2l/tests.log; tail -12 gpurun_out/r2l/tests.log
