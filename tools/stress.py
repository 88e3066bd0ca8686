#!/usr/bin/env python3
"""Randomised GPU-vs-oracle stress: python tools/stress.py [seconds] [first_seed]
Environment: SHARD=n (n virtual ranks), N2=1 (synteny blocks + GlueStripes + reports as well), STAGES=3 (three-stage cascades),
MANY=1 (30 - 70 strains of a few kbp: ids with dozens of instances, mark lists and AnyBulges tables in the arena),
LONGK=1 (vertex sizes 33 .. 1500: the fingerprint path of longk_fp.hip, cascades k -> 2k),
PARKY=1 (a dozen strains, k 15 - 20, 3 - 8 % SNPs: transactions of several collapses side by side -- where parked transactions meet neighbours).
A bounded run of the same loop is part of the GPU suite (tests/test_gpu_stress.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sibelia_amd import BlockFinder, workloads as W        # noqa: E402
from oracle.oracle import Oracle                            # noqa: E402


def draw_case(seed, many=False, stages3=False):
    """the case of `seed`: (sequences, stages, rng, n, L0, snp) -- rng is left where the optional draws of run() continue"""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 13))
    L0 = int(rng.integers(3_000, 80_000))
    if many:
        n, L0 = int(rng.integers(30, 71)), int(rng.integers(2_000, 9_000))
    k = int(rng.choice([15, 16, 20, 25, 31, 32, 40]))
    D = int(rng.integers(k, 12 * k))
    snp = float(rng.choice([0.002, 0.01, 0.03, 0.08]))
    seqs = W.gen_strains(L0=L0, n=n, seed=seed, snp=snp, indel_every=int(rng.choice([200, 1000, 2000])),
                         inv_min=max(50, L0 // 100), inv_max=max(200, L0 // 20))
    stages = [(k, D)] if rng.random() < 0.6 else [(k, D), (int(min(40, k + 5)), D + 50)]
    if os.environ.get("PARKY"):                                 # multi-collapse transactions side by side: small k, many SNPs, a dozen strains (seed 93194's kind)
        n, L0 = int(rng.integers(8, 14)), int(rng.integers(15_000, 60_000))
        k = int(rng.choice([15, 16, 20])); D = int(rng.integers(3 * k, 8 * k)); snp = float(rng.choice([0.03, 0.08]))
        seqs = W.gen_strains(L0=L0, n=n, seed=seed, snp=snp, indel_every=int(rng.choice([200, 1000])), inv_min=max(50, L0 // 100), inv_max=max(200, L0 // 20))
        stages = [(k, D)]
    if os.environ.get("LONGK"):                                 # long vertex sizes: k > 32 in every stage
        k = int(rng.choice([33, 40, 64, 100, 127, 128, 200, 333, 512, 700, 1024, 1500]))
        D = int(rng.integers(k + 1, 4 * k + 50))
        stages = [(k, D)] if rng.random() < 0.5 else [(k, D), (2 * k, 2 * D)]
    if stages3:                                                 # every case is a three-stage cascade (state carried across copy-backs)
        stages = [(k, D), (int(min(40, k + 5)), D + 50), (int(min(48, k + 10)), D + 100)]
    return seqs, stages, rng, n, L0, snp


# a case whose GPU time is more than SLOW_FACTOR x the single-threaded oracle's (and more than SLOW_FLOOR seconds: launch overheads dominate
# tiny cases) is reported like a mismatch: exact-but-pathological runs -- round 5's seed 67000 took 238 s against 4.2 s -- must not pass
SLOW_FACTOR, SLOW_FLOOR = 20.0, 3.0


def run(budget=300.0, seed=1000, stages3=False, nshard=0, n2=False, log=print, count=None):
    """draws cases from `seed` on for `budget` seconds -- or exactly `count` cases when given (the GPU suite: no wall-clock dependence);
    returns (cases, mismatching seeds)"""
    t0 = time.time()
    done, bad = 0, []
    while (done < count) if count is not None else (time.time() - t0 < budget):
        seqs, stages, rng, n, L0, snp = draw_case(seed, bool(os.environ.get("MANY")), stages3)
        head = "case %d n %d L0 %d stages %s snp %s" % (seed, n, L0, stages, snp)
        if nshard:                                                  # the same run through nshard virtual ranks (sharded enumeration)
            from sibelia_amd.dist import LocalShardedFinder
            bf = LocalShardedFinder(seqs, [0] * nshard)
        else:
            bf = BlockFinder(seqs, device=0)
        orc = Oracle(seqs)
        tg = tc = 0.0
        if rng.random() < 0.3 and not nshard:
            # a pinned commit window (results must not depend on it).  MANY=1: not the windows of 1 and 3 entries -- one or three ids per round on
            # 15 000 ids x 4 iterations is 17 734 rounds of 13 ms: round 5's "livelock" of seed 67000 was this knob, not the product (r06.md 1b)
            w = int(rng.choice([1, 3, 64, 1000]))
            bf.set_window(w if not os.environ.get("MANY") else max(w, 64))
        ok = True
        for kk, dd in stages:
            t1 = time.time(); a = bf.simplify_stage(kk, dd, 4); t2 = time.time(); b = orc.simplify_stage(kk, dd, 4); t3 = time.time()
            tg += t2 - t1; tc += t3 - t2
            (sa, pa), (sb, pb) = bf.state(), orc.state()
            ok = ok and a == b and sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
        extra = ""
        if ok and not nshard and n2:                                # synteny blocks + GlueStripes + report texts after the stages
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
            extra = " blocks(%d,%d,%d,%d) %d -> %d" % (bk, tk, ms, sh, len(ga), len(pa_))
        st = bf.stats()
        if nshard:
            st = st[0]
        done += 1
        if not ok:
            bad.append(seed)
            log(head + extra + " MISMATCH")
        elif tg > SLOW_FLOOR and tg > SLOW_FACTOR * tc:
            bad.append(seed)
            log(head + extra + " SLOW bulges %d rounds %d replays %d gpu %.2fs cpu %.2fs (more than %.0f x the oracle)" % (a, st["rounds"], st["replays"], tg, tc, SLOW_FACTOR))
        else:
            log(head + extra + " ok bulges %d rounds %d replays %d gpu %.2fs cpu %.2fs" % (a, st["rounds"], st["replays"], tg, tc))
        bf.close()
        seed += 1
    return done, bad


if __name__ == "__main__":
    done, bad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1000,
                    os.environ.get("STAGES") == "3", int(os.environ.get("SHARD", "0")), bool(os.environ.get("N2")),
                    log=lambda m: print(m, flush=True))
    print("done", done, "mismatches", len(bad), bad)
