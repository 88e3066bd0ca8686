#!/bin/bash
# The round's LAST build: config 3 / 5 kernel tables and bench lines again (longk_fp.hip changed after tools/collect_r06.sh ran; commit.hip
# and its headers did not: the PMC / SQ passes of the default workload stand, profiles/pmc_latest.json carries their digest), then the
# default bench line with the reference on the full workload beside it.
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/c3stats $out/c5stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3stats -- python $R/bench.py --config 3 --no-cpu-baseline > $out/c3stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5stats -- python $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline > $out/c5stats.log 2>&1
find $out -name '*_kernel_trace.csv' -size +20M -delete
cd $R
timeout 600 python bench.py --config 3 2>/dev/null | grep '^{' > $out/bench_config3.json
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 2>/dev/null | grep '^{' > $out/bench_config5.json
timeout 600 python tools/longk_profile.py 100 500 > $out/longk_enumerate.jsonl 2>/dev/null
timeout 1500 python bench.py 2>/dev/null | grep '^{' > $out/bench_default.json
tail -c 600 $out/bench_default.json
