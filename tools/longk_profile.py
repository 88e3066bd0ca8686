#!/usr/bin/env python3
"""Long-k enumeration on the bench workload (8 x 4.6 Mbp by default) for rocprofv3: python tools/longk_profile.py [k ...]
prints the wall time of each call; run under `rocprofv3 --kernel-trace --stats` for the per-kernel table."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sibelia_amd import BlockFinder, workloads as W          # noqa: E402

ks = [int(x) for x in sys.argv[1:]] or [100, 500]
seqs = W.gen_strains(L0=int(os.environ.get("L0", 4_600_000)), n=int(os.environ.get("STRAINS", 8)), seed=1)
bf = BlockFinder(seqs, device=0)
for k in ks:
    bf.enumerate(k)                      # (allocates the workspaces)
    t = time.perf_counter(); r = bf.enumerate(k); dt = time.perf_counter() - t
    print(json.dumps({"k": k, "bif": int(r[0]), "enumerate_call_ms": round(1e3 * dt, 2), "stats": {x: y for x, y in bf.stats().items() if x.endswith("_ms")}}), flush=True)
bf.close()
