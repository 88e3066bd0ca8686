#!/bin/bash
# round 6, fifth session: the fingerprint path measured -- enumeration times at k = 100 / 500 against the doubling, configs 3 and 5, kernel trace
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r6e
mkdir -p $out
cd $R
timeout 600 python tools/longk_profile.py 100 500 > $out/longk_fp.jsonl 2>/dev/null; cat $out/longk_fp.jsonl | cut -c1-400
SBL_LONGK_DOUBLING=1 timeout 600 python tools/longk_profile.py 100 500 > $out/longk_dbl.jsonl 2>/dev/null; cat $out/longk_dbl.jsonl | cut -c1-400
timeout 600 python bench.py --config 3 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_config3.json; python -c "
import json; d=json.loads(open('$out/bench_config3.json').read()); print('config3', d['ms_per_step'], d.get('matches_reference_fixture'), [ (s.get('k'), round(s.get('enumerate_ms',0),2), round(s.get('total_ms',0),2)) for s in d.get('stages',[])] if 'stages' in d else list(d.keys()))"
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_config5.json; python -c "
import json; d=json.loads(open('$out/bench_config5.json').read()); print('config5', d['ms_per_step'], d.get('matches_reference_fixture'), d.get('phase_ms'))"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3stats -- python $R/bench.py --config 3 --no-cpu-baseline > $out/c3stats.log 2>&1
find $out -name '*_kernel_trace.csv' -size +20M -delete
f=$(find $out/c3stats -name '*kernel_stats.csv' | head -1); head -25 "$f" | cut -c1-200
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5stats -- python $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline > $out/c5stats.log 2>&1
find $out -name '*_kernel_trace.csv' -size +20M -delete
f=$(find $out/c5stats -name '*kernel_stats.csv' | head -1); head -22 "$f" | cut -c1-160
