#!/bin/bash
# round 6, first session: parking everywhere (chain + parking, three virtual ranks), the resume list, a quick bench line
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r6a
mkdir -p $out
cd $R
timeout 1200 python -m pytest tests/test_gpu_parking.py -m gpu -q --timeout=900 -k "not three_virtual" > $out/park.log 2>&1; echo "rc $?" >> $out/park.log; tail -5 $out/park.log
SBL_TRACE=1 timeout 600 python -m pytest tests/test_gpu_parking.py -m gpu -q -x --timeout=500 -k "three_virtual" > $out/park3.log 2>&1; echo "rc $?" >> $out/park3.log; tail -25 $out/park3.log | cut -c1-400
for envs in "" "SBL_PARK=0"; do
  env $envs timeout 600 python bench.py --no-cpu-baseline --steps 5 > $out/bench.json 2> $out/bench.err
  python - "$out/bench.json" "[$envs]" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"], 2), "rounds", d["config"]["rounds"], "replays", d["config"]["replays"], "match", d["matches_reference_fixture"], {k: round(v, 2) for k, v in d["phase_ms"].items() if k in ("probe_ms", "reserve_ms", "commit_ms", "snapshot_ms", "enumerate_ms", "simplify_ms")})
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
MANY=1 timeout 300 python tools/stress.py 120 67000 > $out/many.log 2>&1; tail -4 $out/many.log
