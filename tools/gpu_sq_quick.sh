#!/bin/bash
# SQ counters of the round kernels (through gpurun): usage tools/gpu_sq_quick.sh <tag> [env settings]
tag=${1:-sq}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/sq.log 2>&1
cd $R
python tools/sq_summary.py _tmp_$tag $out/sq | cut -c1-400
mv profiles/_tmp_${tag}_sq_counters.json $out/sq_counters.json
find $out -name '*_kernel_trace.csv' -size +20M -delete; find $out -name '*_counter_collection.csv' -size +20M -delete
