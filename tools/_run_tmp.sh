cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -x -q -k "snp_k5 or short_chr or ambig or strains4_100k_fine" > gpurun_out/r2n/graph.log 2>&1; tail -6 gpurun_out/r2n/graph.log
