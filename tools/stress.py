#!/usr/bin/env python3
"""Randomised GPU-vs-oracle stress (not part of the test suite): python tools/stress.py [seconds] [first_seed]
Environment: SHARD=n (n virtual ranks), N2=1 (synteny blocks + GlueStripes + reports as well), STAGES=3 (three-stage cascades)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sibelia_amd import BlockFinder, workloads as W        # noqa: E402
from oracle.oracle import Oracle                            # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0 = time.time()
done = bad = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 13))
    L0 = int(rng.integers(3_000, 80_000))
    k = int(rng.choice([15, 16, 20, 25, 31, 32, 40]))
    D = int(rng.integers(k, 12 * k))
    snp = float(rng.choice([0.002, 0.01, 0.03, 0.08]))
    seqs = W.gen_strains(L0=L0, n=n, seed=seed, snp=snp, indel_every=int(rng.choice([200, 1000, 2000])),
                         inv_min=max(50, L0 // 100), inv_max=max(200, L0 // 20))
    stages = [(k, D)] if rng.random() < 0.6 else [(k, D), (int(min(40, k + 5)), D + 50)]
    if __import__("os").environ.get("STAGES") == "3":                  # STAGES=3: every case is a three-stage cascade (state carried across copy-backs)
        stages = [(k, D), (int(min(40, k + 5)), D + 50), (int(min(48, k + 10)), D + 100)]
    print("case", seed, "n", n, "L0", L0, "stages", stages, "snp", snp, end=" ", flush=True)
    nshard = int(__import__("os").environ.get("SHARD", "0"))          # SHARD=3: the same run through 3 virtual ranks (sharded enumeration)
    if nshard:
        from sibelia_amd.dist import LocalShardedFinder
        bf = LocalShardedFinder(seqs, [0] * nshard)
    else:
        bf = BlockFinder(seqs, device=0)
    orc = Oracle(seqs)
    tg = tc = 0.0
    if rng.random() < 0.3 and not nshard:
        bf.set_window(int(rng.choice([1, 3, 64, 1000])))
    ok = True
    for kk, dd in stages:
        t1 = time.time(); a = bf.simplify_stage(kk, dd, 4); t2 = time.time(); b = orc.simplify_stage(kk, dd, 4); t3 = time.time()
        tg += t2 - t1; tc += t3 - t2
        (sa, pa), (sb, pb) = bf.state(), orc.state()
        ok = ok and a == b and sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    if ok and not nshard and __import__("os").environ.get("N2"):      # N2=1: synteny blocks + GlueStripes + report texts after the stages (and before them: raw graph)
        bk = int(rng.choice([stages[-1][0], max(4, stages[-1][0] // 2), 2 * stages[-1][0]]))
        tk = int(min(bk, rng.choice([bk, max(3, bk // 2), 30])))
        ms = int(rng.choice([bk, 50, 500]))
        sh = bool(rng.random() < 0.2)
        ga, gb = bf.generate_blocks(bk, tk, ms, sh), orc.generate_blocks(bk, tk, ms, sh)
        ok = len(ga) == len(gb) and all((ga[f] == gb[f]).all() for f in ("id", "chr", "start", "end"))
        names = ["s%d" % i for i in range(n)]
        pa_, ta_ = bf.postprocess(names, True)
        pb_, tb_ = orc.postprocess(gb, names, True)
        ok = ok and len(pa_) == len(pb_) and all((pa_[f] == pb_[f]).all() for f in ("id", "chr", "start", "end")) and list(ta_) == list(tb_)
        print("blocks(%d,%d,%d,%d) %d -> %d" % (bk, tk, ms, sh, len(ga), len(pa_)), end=" ")
    st = bf.stats()
    if nshard: st = st[0]
    done += 1
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "n", n, "L0", L0, "stages", stages, "snp", snp, flush=True)
    else:
        print("ok bulges", a, "rounds", st["rounds"], "replays", st["replays"], "gpu %.2fs cpu %.2fs" % (tg, tc), flush=True)
    bf.close()
    seed += 1
print("done", done, "mismatches", bad)
