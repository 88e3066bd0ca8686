#!/usr/bin/env python3
"""Replays one golden vector on the GPU with timing: python tools/one_vector.py small/002"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests import vectors as V
from sibelia_amd import BlockFinder
v = [x for x in V.load_vectors() if x["name"] == sys.argv[1]][0]
seqs = V.vector_input(v)
bf = BlockFinder(seqs, device=0)
for o in v["outputs"]:
    t = time.time(); got = V.run_cmd(bf, o["cmd"]); dt = time.time() - t
    st = bf.stats()
    print(o["cmd"], "ok" if V.F.sha256(got) == o["sha256"] else "MISMATCH", "%.2fs" % dt,
          {k: st[k] for k in ("bif_count", "instances", "bulges", "rounds", "transactions", "chain_transactions")} if o["cmd"].startswith("stage") else "", flush=True)
