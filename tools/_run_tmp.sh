cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not densest and not full_size" > gpurun_out/r2d/tests.log 2>&1; tail -3 gpurun_out/r2d/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2d/bench.log 2>&1; grep '^{' gpurun_out/r2d/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_ms'], d['config']['bulges'])" || tail -20 gpurun_out/r2d/bench.log
