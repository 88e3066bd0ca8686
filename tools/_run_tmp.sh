cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not dense and not densest and not full_size_stage" > gpurun_out/r2b/tests.log 2>&1; tail -5 gpurun_out/r2b/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2b/bench.log 2>&1; grep '^{' gpurun_out/r2b/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_ms'])" || tail -20 gpurun_out/r2b/bench.log
