#!/bin/bash
# round 6: the slow many-strains cases (SBL_DENSE_MAX_ELEMS=2000000 in front of the command: straight through the one-launch path)
cd ${GRAFT_REPO_ROOT:-$PWD}
cat > /tmp/slowcase.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import stress
from sibelia_amd import BlockFinder
from oracle.oracle import Oracle
import numpy as np
seed = int(sys.argv[1])
seqs, stages, *_ = stress.draw_case(seed, True)
bf = BlockFinder(seqs, device=0); orc = Oracle(seqs)
for kk, dd in stages[:1]:
    t0 = time.time(); n = bf.simplify_stage(kk, dd, 4); tg = time.time() - t0
    t0 = time.time(); m = orc.simplify_stage(kk, dd, 4); tc = time.time() - t0
    (sa, pa), (sb, pb) = bf.state(), orc.state()
    ok = n == m and sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    st = bf.stats()
    print("seed", seed, "DENSE_MAX", os.environ.get("SBL_DENSE_MAX_ELEMS"), "ok", ok, "bulges", n, "rounds", st["rounds"], "gpu %.2f s cpu %.2f s" % (tg, tc), flush=True)
PY
for s in 67008 67012 67000; do
  timeout 250 python /tmp/slowcase.py $s 2>&1 | tail -1
done
