cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_fasta.py -x -q > gpurun_out/r2c/fasta.log 2>&1; tail -25 gpurun_out/r2c/fasta.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not dense and not densest and not full_size" > gpurun_out/r2c/tests.log 2>&1; tail -3 gpurun_out/r2c/tests.log
