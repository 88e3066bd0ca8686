#!/bin/bash
# round 6: the reservation's certificate / light first bursts: parity, then A/B (SBL_TEST_FLAGS=131072: no certificates)
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r6i
mkdir -p $out
cd $R
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parking.py tests/test_gpu_block_index.py -m gpu -q -x --timeout=500 -k "not config5_full and not config4 and not config3_full" > $out/tests.log 2>&1; tail -3 $out/tests.log | cut -c1-300
for envs in "" "SBL_TEST_FLAGS=131072" ""; do
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
