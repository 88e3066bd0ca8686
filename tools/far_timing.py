#!/usr/bin/env python3
"""Per-stage timing of a parameter set on one of the example inputs (GPU box): python tools/far_timing.py [far|fine|loose] [input]"""
import gzip
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sibelia_amd import BlockFinder      # noqa: E402

SETS = {"loose": [(30, 150), (100, 1000), (1000, 5000), (5000, 15000)], "far": [(15, 120), (100, 500), (500, 1500)],
        "fine": [(30, 150), (100, 500), (500, 1500)]}
name = sys.argv[1] if len(sys.argv) > 1 else "far"
inp = sys.argv[2] if len(sys.argv) > 2 else "Helicobacter_pylori"
with tempfile.TemporaryDirectory() as d:
    fa = os.path.join(d, "in.fa")
    open(fa, "wb").write(gzip.open(os.path.join(ROOT, "tests", "golden", "data", inp + ".fa.gz")).read())
    bf = BlockFinder.from_fasta(fa, device=0)
for k, D in SETS[name]:
    t0 = time.time()
    b = bf.PerformGraphSimplifications(k, D, 4)
    st = bf.stats()
    print("stage k=%d D=%d: %.2f s, bulges %d" % (k, D, time.time() - t0, b),
          {x: (round(st[x], 1) if isinstance(st[x], float) else st[x]) for x in ("bif_count", "instances", "iterations", "rounds", "replays", "grow_replays", "transactions", "chain_transactions",
                                                                                  "enumerate_ms", "snapshot_ms", "probe_ms", "reserve_ms", "commit_ms", "simplify_ms")}, flush=True)
