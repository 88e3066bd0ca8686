#!/bin/bash
# round 6: stress of the final build -- default mode, long k, three-stage cascades, blocks + reports, three virtual ranks (long k too)
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r6g
mkdir -p $out
cd $R
export SBL_CHECK_INDEX=1
timeout 200 python tools/stress.py 150 920000 > $out/default.log 2>&1; tail -1 $out/default.log
LONGK=1 timeout 200 python tools/stress.py 150 930000 > $out/longk.log 2>&1; tail -1 $out/longk.log
LONGK=1 SHARD=3 timeout 200 python tools/stress.py 120 940000 > $out/longk_shard.log 2>&1; tail -1 $out/longk_shard.log
STAGES=3 timeout 150 python tools/stress.py 100 950000 > $out/stages3.log 2>&1; tail -1 $out/stages3.log
N2=1 timeout 150 python tools/stress.py 100 960000 > $out/n2.log 2>&1; tail -1 $out/n2.log
SHARD=3 timeout 150 python tools/stress.py 100 970000 > $out/shard3.log 2>&1; tail -1 $out/shard3.log
grep -h "SLOW\|MISMATCH" $out/*.log | head -10 | cut -c1-250
