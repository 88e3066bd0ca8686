cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bucket_overflow" > gpurun_out/r2m/ovf.log 2>&1; tail -12 gpurun_out/r2m/ovf.log
