#!/usr/bin/env python3
"""Times the dense-conflict golden vectors on the GPU, smallest first: python tools/dense_vectors.py [budget_s] [min_bulges]"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests import vectors as V
from sibelia_amd import BlockFinder
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
minb = int(sys.argv[2]) if len(sys.argv) > 2 else 400
vs = V.load_vectors()
bul = lambda v: sum(o.get("bulges", 0) for o in v["outputs"])
dense = sorted([v for v in vs if v["name"].startswith("small/") and bul(v) >= minb], key=bul)
t0 = time.time()
for v in dense:
    if time.time() - t0 > budget:
        print("budget exhausted before", v["name"], bul(v)); break
    seqs = V.vector_input(v)
    print(v["name"], "bulges", bul(v), "records", len(seqs), "bp", sum(map(len, seqs)), [o["cmd"] for o in v["outputs"]], end=" ", flush=True)
    t = time.time()
    bf = BlockFinder(seqs, device=0)
    ok = True
    tot = {"rounds": 0, "chain_transactions": 0, "transactions": 0, "replays": 0}
    for o in v["outputs"]:
        got = V.run_cmd(bf, o["cmd"])
        ok = ok and V.F.sha256(got) == o["sha256"]
        if o["cmd"].startswith("stage"):
            st = bf.stats()
            for kk in tot: tot[kk] += st[kk]
    print("ok" if ok else "MISMATCH", "%.2fs" % (time.time() - t), tot, flush=True)
    bf.close()
