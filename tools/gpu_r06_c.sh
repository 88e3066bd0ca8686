#!/bin/bash
# round 6, third session: the parking suite, the full seed-67000 case with and without parking (for the record), sharded + many-strains stress
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r6c
mkdir -p $out
cd $R
timeout 1500 python -m pytest tests/test_gpu_parking.py tests/test_gpu_perf_guard.py -m gpu -q --timeout=900 > $out/park.log 2>&1; echo "rc $?" >> $out/park.log; tail -6 $out/park.log | cut -c1-300
cat > /tmp/case67000.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import stress
from sibelia_amd import BlockFinder
seqs, stages, *_ = stress.draw_case(67000, True)
bf = BlockFinder(seqs, device=0)
t0 = time.time()
n = bf.simplify_stage(*stages[0], 4)
st = bf.stats()
print("SBL_PARK=%s" % os.environ.get("SBL_PARK", "default"), "bulges", n, "rounds", st["rounds"], "seconds", round(time.time() - t0, 2), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items() if k in ("transactions", "chain_transactions", "executed", "iterations", "replays", "probe_ms", "reserve_ms", "commit_ms", "snapshot_ms", "simplify_ms", "enumerate_ms")})
PY
SBL_PARK=0 timeout 400 python /tmp/case67000.py > $out/c67000_park0.txt 2>&1; tail -1 $out/c67000_park0.txt | cut -c1-500
timeout 400 python /tmp/case67000.py > $out/c67000_default.txt 2>&1; tail -1 $out/c67000_default.txt | cut -c1-500
SHARD=3 timeout 200 python tools/stress.py 90 7100 > $out/shard3.log 2>&1; tail -1 $out/shard3.log
MANY=1 timeout 300 python tools/stress.py 150 67001 > $out/many.log 2>&1; tail -3 $out/many.log | cut -c1-300
