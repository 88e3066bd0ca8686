cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 1500 python bench.py --k 15 --D 120 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2k/bench_far.log 2>&1; grep '^{' gpurun_out/r2k/bench_far.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms'], d['config'])" || tail -5 gpurun_out/r2k/bench_far.log
