cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests/test_gpu_shard.py -x -q --durations=5 > gpurun_out/r2i/shard.log 2>&1; tail -12 gpurun_out/r2i/shard.log
timeout 600 python bench.py --no-cpu-baseline --shard-enum > gpurun_out/r2i/bench_shard1.log 2>&1; grep '^{' gpurun_out/r2i/bench_shard1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_ms'], d['config']['bulges'], d['config']['exchange_bytes_rank0'])" || tail gpurun_out/r2i/bench_shard1.log
timeout 300 python tools/shard_probe.py > gpurun_out/r2i/shard_probe.log 2>&1; tail -6 gpurun_out/r2i/shard_probe.log
