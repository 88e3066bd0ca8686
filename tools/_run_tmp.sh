cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 1200 python -m pytest tests/test_gpu_shard.py -x -q -k "rccl" --durations=5 > gpurun_out/r2f/rccl.log 2>&1; tail -15 gpurun_out/r2f/rccl.log
