"""Small stage run for debugging the plan / execute split: GPU result against the oracle, with the round trace."""
import sys, time
import numpy as np
from sibelia_amd import BlockFinder, workloads as W
from oracle.oracle import Oracle
L0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
seqs = W.gen_strains(L0=L0, n=4, seed=9, inv_min=2000, inv_max=8000)
bf, orc = BlockFinder(seqs, device=0), Oracle(seqs)
t = time.time()
a = bf.simplify_stage(25, 150, 4)
print("gpu", a, round(time.time() - t, 2), flush=True)
b = orc.simplify_stage(25, 150, 4)
(sa, pa), (sb, pb) = bf.state(), orc.state()
print("oracle", b, "match", a == b and sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb)))
