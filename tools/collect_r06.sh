#!/bin/bash
# Runs on the GPU box (through gpurun): everything profiles/r06_* is made of -> gpurun_out/prof_<tag>/ (tools/profile_summary.py,
# tools/sq_summary.py turn it into the committed summaries afterwards, in the build container)
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# ---- default workload: kernel stats, two PMC passes, SQ counters
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --no-cpu-baseline > $out/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/sq.log 2>&1
# ---- config 3 (8 strains, -s fine cascade): kernel stats + the two PMC passes
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3stats -- python $R/bench.py --config 3 --no-cpu-baseline > $out/c3stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/c3fetch -- python $R/bench.py --config 3 --steps 1 --warmup 0 --no-cpu-baseline > $out/c3fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/c3write -- python $R/bench.py --config 3 --steps 1 --warmup 0 --no-cpu-baseline > $out/c3write.log 2>&1
find $out -name '*_kernel_trace.csv' -size +20M -delete
cd $R
# ---- the bench lines: default (with the reference on the full workload beside it), configs 3 / 4 / 5, the round trace, the RCCL self-test
SBL_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/trace_bench.json 2> $out/trace.err
timeout 600 python bench.py --config 3 2>/dev/null | grep '^{' > $out/bench_config3.json
timeout 900 python bench.py --config 4 --steps 2 --warmup 1 2>/dev/null | grep '^{' > $out/bench_config4.json
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 2>/dev/null | grep '^{' > $out/bench_config5.json
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5stats -- python $R/bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline > $out/c5stats.log 2>&1; find $out -name '*_kernel_trace.csv' -size +20M -delete; cd $R
timeout 300 python bench.py --gpus 2 --dry-collectives 2>/dev/null | grep '^{' > $out/rccl_two_ranks_one_gpu.json
timeout 600 python tools/longk_profile.py 100 500 > $out/longk_enumerate.jsonl 2>/dev/null
SBL_LONGK_DOUBLING=1 timeout 600 python tools/longk_profile.py 100 500 > $out/longk_enumerate_doubling.jsonl 2>/dev/null
# (the default line last -- and, for the round's LAST build, by tools/collect_r06_final.sh bench in a second call, with profiles/pmc_latest.json of this build in place: SKIP_DEFAULT=1)
[ -n "$SKIP_DEFAULT" ] || timeout 1500 python bench.py 2>/dev/null | grep '^{' > $out/bench_default.json
ls -la $out | head -40; tail -c 500 $out/bench_default.json
