#!/bin/bash
# The short form of tools/collect_r05.sh for the round's LAST build: kernel stats, the two PMC passes and the SQ counters of the default
# workload, then -- with profiles/pmc_latest.json of THIS build in place (second gpurun call) -- the bench line.  usage: collect_r05_final.sh pmc|bench
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_r05
mkdir -p $out
if [ "$1" = "pmc" ]; then
  rm -rf $out/stats $out/fetch $out/write $out/sq
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --no-cpu-baseline > $out/stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/write.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/sq.log 2>&1
  find $out -name '*_kernel_trace.csv' -size +20M -delete
else
  cd $R
  timeout 1500 python bench.py 2>/dev/null | grep '^{' > $out/bench_default.json
  tail -c 300 $out/bench_default.json
fi
