#!/bin/bash
# One GPU-box session (through gpurun): a pytest selection first, then the default bench line under several environment settings
# (A/B of measurement switches on the same box).  usage: tools/gpu_ab_env.sh <tag> "<pytest args or empty>" "ENV1=.. ENV2=.." "ENV3=.." ...
tag=${1:-ab}; shift
sel=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
if [ -n "$sel" ]; then timeout 1500 python -m pytest $sel -m gpu -q --timeout=900 -x > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log; tail -15 $out/tests.log; fi
i=0
for envs in "" "$@"; do
  i=$((i+1))
  env $envs timeout 600 python bench.py --no-cpu-baseline --steps ${STEPS:-5} ${BENCH_ARGS} > $out/bench_$i.json 2> $out/bench_$i.err
  python - "$out/bench_$i.json" "[$envs]" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"], 2), "rounds", d["config"]["rounds"], "replays", d["config"]["replays"], "match", d["matches_reference_fixture"], {k: round(v, 2) for k, v in d["phase_ms"].items() if k in ("probe_ms", "reserve_ms", "commit_ms", "snapshot_ms", "enumerate_ms", "simplify_ms")})
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
