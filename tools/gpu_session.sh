#!/bin/bash
# One GPU-box session (through gpurun): the new / risky tests first (fast feedback), then the whole GPU suite, then the bench line.
# usage: tools/gpu_session.sh <tag> [pytest -k expression for the first pass]
tag=${1:-s}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
if [ -n "$2" ]; then timeout 900 python -m pytest tests -m gpu -q --timeout=600 -k "$2" > $out/first.log 2>&1; echo "rc $?" >> $out/first.log; tail -30 $out/first.log; fi
if [ -z "$SKIP_SUITE" ]; then timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $out/gpu_tests.log 2>&1; echo "pytest rc $?" >> $out/gpu_tests.log; tail -40 $out/gpu_tests.log; fi
timeout 600 python bench.py --no-cpu-full > $out/bench.log 2> $out/bench.err
grep '^{' $out/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', round(d['ms_per_step'],2), 'value', round(d['value']/1e6,1), 'M/s rounds', d['config']['rounds'], 'match', d['matches_reference_fixture'], {k: round(v,2) for k,v in d['phase_ms'].items()})
"
