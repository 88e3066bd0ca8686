#!/bin/bash
# round 6, second session: three virtual ranks with parking; trace of the seed-67000 case under several park caps
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r6b
mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_gpu_parking.py -m gpu -q -x --timeout=500 -k "three_virtual" > $out/park3.log 2>&1; echo "rc $?" >> $out/park3.log; tail -8 $out/park3.log | cut -c1-300
cat > /tmp/case67000.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import stress
from sibelia_amd import BlockFinder
seqs, stages, *_ = stress.draw_case(67000, True)
bf = BlockFinder(seqs, device=0)
t0 = time.time()
n = bf.simplify_stage(*stages[0], 4)
print("bulges", n, "rounds", bf.stats()["rounds"], "seconds", round(time.time() - t0, 2), {k: v for k, v in bf.stats().items() if k in ("transactions", "chain_transactions", "executed", "iterations", "replays")})
PY
for cap in 0 2 4; do
  SBL_PARK=$cap SBL_TRACE=1 timeout 120 python /tmp/case67000.py > $out/c$cap.out 2> $out/c$cap.err
  echo "== cap $cap: $(tail -1 $out/c$cap.out)"; grep -c "round" $out/c$cap.err; grep "chain mode" $out/c$cap.err | head -5
  grep " round " $out/c$cap.err | awk 'NR<=60 || NR%200==0' | cut -c1-160 > $out/c$cap.sample; rm -f $out/c$cap.err.full; head -c 3000000 $out/c$cap.err > $out/c$cap.trace; rm $out/c$cap.err
done
