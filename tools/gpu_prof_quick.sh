#!/bin/bash
# quick per-kernel profile of the default bench (through gpurun): usage tools/gpu_prof_quick.sh <tag> [env settings]
tag=${1:-pq}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
env "$@" SBL_TEST_FLAGS=32 timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $out/flags.json 2> $out/flags.err
grep 'block index' $out/flags.err | tail -2
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/stats.log 2>&1
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-60s calls %6s total %9.3f ms avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
find $out -name '*_kernel_trace.csv' -size +20M -delete
