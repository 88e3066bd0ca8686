#!/bin/bash
# round 6: the probe's single collective: parity of the affected suites (virtual ranks), then the bench line
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r6f
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_gpu_parking.py tests/test_gpu_shard.py tests/test_gpu_errors.py -m gpu -q -x --timeout=300 > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log; tail -6 $out/tests.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_stress.py -m gpu -q -x --timeout=280 > $out/stress_t.log 2>&1; tail -2 $out/stress_t.log | cut -c1-300
for envs in "" "SBL_PARK=0"; do
  env $envs timeout 200 python bench.py --no-cpu-baseline --steps 8 > $out/bench.json 2> $out/bench.err
  python - "$out/bench.json" "[$envs]" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"], 2), "rounds", d["config"]["rounds"], "replays", d["config"]["replays"], "match", d["matches_reference_fixture"], {k: round(v, 2) for k, v in d["phase_ms"].items() if k in ("probe_ms", "reserve_ms", "commit_ms", "snapshot_ms", "enumerate_ms", "simplify_ms")})
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
