#!/bin/bash
# round 6: ordered rounds against the one-launch path (SBL_DENSE_MAX_ELEMS raised) case by case: where is which faster?
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/r6k; mkdir -p $out; cd $R
cat > /tmp/ab_dense.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import stress
from sibelia_amd import BlockFinder
many = bool(int(sys.argv[1])); first = int(sys.argv[2]); count = int(sys.argv[3])
for seed in range(first, first + count):
    seqs, stages, *_ = stress.draw_case(seed, many)
    E = sum(len(s) for s in seqs) + len(seqs) + 1
    res = []
    for dm in (None, "4000000"):
        if dm: os.environ["SBL_DENSE_MAX_ELEMS"] = dm
        else: os.environ.pop("SBL_DENSE_MAX_ELEMS", None)
        bf = BlockFinder(seqs, device=0)
        bf.simplify_stage(*stages[0], 4)          # warm (allocations)
        bf.close()
        bf = BlockFinder(seqs, device=0)
        t0 = time.time(); n = bf.simplify_stage(*stages[0], 4); dt = time.time() - t0
        st = bf.stats(); bf.close()
        res.append((dt, st["rounds"], n, st["instances"], st["bif_count"]))
    k, D = stages[0]
    L0 = len(seqs[0])
    print("seed %d n %d L0 %d E %d k %d D %d inst/id %.1f par %.1f rounds %d : rounds %.3f s dense %.3f s ratio %.2f" % (seed, len(seqs), L0, E, k, D, res[0][3] / max(1, res[0][4]), L0 / (5.0 * (D + 2 * k)), res[0][1], res[0][0], res[1][0], res[0][0] / max(1e-9, res[1][0])), flush=True)
PY
timeout 400 python /tmp/ab_dense.py 0 980000 40 > $out/default.log 2>&1
timeout 500 python /tmp/ab_dense.py 1 67001 14 > $out/many.log 2>&1
cat $out/default.log | cut -c1-200; cat $out/many.log | cut -c1-200
