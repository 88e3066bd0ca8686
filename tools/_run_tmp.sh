cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2p/tests.log 2>&1; echo "rc $?" >> gpurun_out/r2p/tests.log; tail -6 gpurun_out/r2p/tests.log
bash tools/collect_profiles.sh r02 > gpurun_out/r2p/prof.log 2>&1; tail -c 300 gpurun_out/prof_r02/bench_default.json
