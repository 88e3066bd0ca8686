#!/bin/bash
# Runs on the GPU box (through gpurun): kernel stats + two PMC passes + the default bench line -> gpurun_out/prof_<tag>/
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --no-cpu-baseline > $out/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/write.log 2>&1
cd $R && timeout 900 python bench.py --check --no-cpu-full 2>/dev/null | grep '^{' > $out/bench_default.json
# keep only what the summary needs (the raw traces are large)
find $out -name '*_kernel_trace.csv' -size +20M -delete
ls -la $out $out/*/* | head -40
tail -c 600 $out/bench_default.json
