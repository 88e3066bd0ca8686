cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not densest and not full_size and not config5" > gpurun_out/r2g/tests.log 2>&1; tail -3 gpurun_out/r2g/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2g/bench.log 2>&1; grep '^{' gpurun_out/r2g/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_ms'], d['config']['bulges'], d.get('pcie_inclusive'))" || tail -20 gpurun_out/r2g/bench.log
