#!/usr/bin/env python3
"""Dense-conflict regime probe (small k): python tools/dense_case.py L0 n k D snp"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from sibelia_amd import workloads as W, BlockFinder
from oracle.oracle import Oracle
L0, n, k, D, snp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
seqs = W.gen_strains(L0=L0, n=n, seed=7, snp=snp, indel_every=1000, inv_min=max(50, L0 // 100), inv_max=max(200, L0 // 20))
bf, o = BlockFinder(seqs, device=0), Oracle(seqs)
t = time.time(); b = o.simplify_stage(k, D, 4); tc = time.time() - t
print("oracle bulges", b, "%.1fs" % tc, flush=True)
t = time.time(); a = bf.simplify_stage(k, D, 4); tg = time.time() - t
(sa, pa), (sb, pb) = bf.state(), o.state()
st = bf.stats()
print(L0, n, k, D, 'bulges', a, b, 'equal', a == b and sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb)), 'gpu %.1fs cpu %.1fs' % (tg, tc),
      'ids', st['bif_count'], 'inst', st['instances'], 'rounds', st['rounds'], 'chain', st['chain_transactions'], 'txn', st['transactions'], 'executed', st['executed'], 'replays', st['replays'],
      {x: round(st[x]) for x in ('snapshot_ms', 'probe_ms', 'reserve_ms', 'commit_ms', 'simplify_ms')}, flush=True)
