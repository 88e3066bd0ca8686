#!/bin/bash
# 8-strain (default) and 62-strain bench lines in short form.  usage: tools/bench_pair.sh [env assignments...]
fmt='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), d["config"]["rounds"], d["config"]["bulges"], d["matches_reference_fixture"], d["state_sha256"][:16], {k: round(v,2) for k,v in d["phase_ms"].items() if k in ("commit_ms","probe_ms","reserve_ms","snapshot_ms","enumerate_ms")})'
env "$@" python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "$fmt"
env "$@" python bench.py --strains 62 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "$fmt"
