#!/bin/bash
# The round's LAST build (commit.hip changed once more after tools/collect_r06.sh ran: the one-launch path): the passes of the default
# workload again -- kernel stats, FETCH_SIZE, WRITE_SIZE, SQ counters -- and config 3 / 5 tables and lines.  usage: collect_r06_final.sh pmc|bench
# (bench: the default line with the reference on the full workload beside it, AFTER profiles/pmc_latest.json of this build is in place)
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_r06
mkdir -p $out
if [ "$1" = "pmc" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $out/stats $out/fetch $out/write $out/sq $out/c3stats $out/c5stats
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --no-cpu-baseline > $out/stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/write.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/sq.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3stats -- python $R/bench.py --config 3 --no-cpu-baseline > $out/c3stats.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5stats -- python $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline > $out/c5stats.log 2>&1
  find $out -name '*_kernel_trace.csv' -size +20M -delete
  cd $R
  timeout 600 python bench.py --config 3 2>/dev/null | grep '^{' > $out/bench_config3.json
  timeout 900 python bench.py --config 4 --steps 2 --warmup 1 2>/dev/null | grep '^{' > $out/bench_config4.json
  timeout 900 python bench.py --config 5 --steps 2 --warmup 1 2>/dev/null | grep '^{' > $out/bench_config5.json
  ls $out
else
  cd $R
  timeout 1500 python bench.py 2>/dev/null | grep '^{' > $out/bench_default.json
  tail -c 400 $out/bench_default.json
fi
