#!/bin/bash
# Runs on the GPU box (through gpurun): everything profiles/r04_* is made of -> gpurun_out/prof_<tag>/
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
bash $R/tools/collect_profiles.sh $tag > $out/collect.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/sq.log 2>&1
find $out -name '*_kernel_trace.csv' -size +20M -delete
cd $R
SBL_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/trace_bench.json 2> $out/trace.err
timeout 600 python bench.py --strains 62 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_62strains.json
timeout 600 python tools/longk_shard_probe.py 4600000 8 100 500 > $out/longk_shard_probe.jsonl 2> $out/longk_shard_probe.err
timeout 600 python tools/longk_shard_probe.py 225000000 0 5000 > $out/longk_shard_probe_config5.jsonl 2>> $out/longk_shard_probe.err
timeout 600 python tools/big_longk.py 900000000 4 5000 15000 > $out/config5_timing.txt 2>&1
ls -la $out; tail -c 400 $out/bench_default.json; echo; cat $out/longk_shard_probe.jsonl | cut -c1-300; cat $out/config5_timing.txt | tail -2 | cut -c1-300
