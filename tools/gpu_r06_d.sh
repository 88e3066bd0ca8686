#!/bin/bash
# round 6, fourth session: the fingerprint path for k > 32 (tests), perf guard, phase profile of two slow many-strains cases
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r6d
mkdir -p $out
cd $R
timeout 1200 python -m pytest tests/test_gpu_longk_fp.py -m gpu -q -x --timeout=600 > $out/fp.log 2>&1; echo "rc $?" >> $out/fp.log; tail -30 $out/fp.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_perf_guard.py -m gpu -q --timeout=500 > $out/guard.log 2>&1; tail -3 $out/guard.log | cut -c1-300
cat > /tmp/slowcase.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import stress
from sibelia_amd import BlockFinder
seed = int(sys.argv[1])
seqs, stages, *_ = stress.draw_case(seed, True)
bf = BlockFinder(seqs, device=0)
for kk, dd in stages[:1]:
    t0 = time.time()
    n = bf.simplify_stage(kk, dd, 4)
    st = bf.stats()
    print("seed", seed, "stage", kk, dd, "bulges", n, "rounds", st["rounds"], "seconds", round(time.time() - t0, 2), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items() if k in ("instances", "bif_count", "transactions", "chain_transactions", "executed", "iterations", "replays", "probe_ms", "reserve_ms", "commit_ms", "snapshot_ms", "simplify_ms")})
PY
SBL_PHASES=1 SBL_TRACE=1 timeout 200 python /tmp/slowcase.py 67008 > $out/s67008.out 2> $out/s67008.err; tail -2 $out/s67008.out | cut -c1-600; grep -v " round \|violation" $out/s67008.err | head -60 | cut -c1-400 > $out/s67008.prof; head -70 $out/s67008.prof; grep " round " $out/s67008.err | head -70 | cut -c1-200 > $out/s67008.rounds; rm $out/s67008.err
