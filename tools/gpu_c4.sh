#!/bin/bash
# 62-strain session (through gpurun): the many-instances parity tests, then bench.py --config 4 under the given environment settings.
# usage: tools/gpu_c4.sh <tag> "ENV=.." ...
tag=${1:-c4}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=1200 -x -k "62 or many or config4 or strains" > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log; grep -E "passed|failed|rc" $out/tests.log | cut -c1-200
i=0
for envs in "" "$@"; do
  i=$((i+1))
  env $envs timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_$i.json 2> $out/bench_$i.err
  python - "$out/bench_$i.json" "[$envs]" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"], 1), "rounds", d["config"].get("rounds"), {k: round(v, 1) for k, v in d.get("phase_ms", {}).items() if k in ("probe_ms", "reserve_ms", "commit_ms", "snapshot_ms", "enumerate_ms", "simplify_ms")}, d.get("state_sha256", "")[:12])
except Exception as e:
    print(sys.argv[2], "no bench line:", e)
PY
done
