#!/usr/bin/env python3
"""One case of tools/stress.py in detail: python tools/diag_case.py SEED -- bulge counts of both sides, where the states differ, the statistics.
Environment: MANY / STAGES as for tools/stress.py; NOWIN=1 ignores the case's pinned commit window; SBL_DBG_LIB=<library of
tools/build_dbg.sh> + DBG_OUT=<file.npy>: also writes, per id, the collapses reported / the first park (round << 8 | collapses) / the
first finish (round << 8 | resumed << 7 | collapses), and <file>_log.npy with one row per collapse (id, round, collapse number,
(source slot << 1) | strand, (target slot << 1) | strand, dS, dT, resumed) -- two runs under different switches are compared offline."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stress                                               # noqa: E402
from sibelia_amd import BlockFinder                         # noqa: E402
from oracle.oracle import Oracle                            # noqa: E402

seed = int(sys.argv[1])
DBG = os.environ.get("SBL_DBG_LIB")                        # a library built with -DSBL_DBG_IDRET (commit.hip): collapses and parks per id
if DBG:
    import sibelia_amd.api as _api
    _api.LIB = DBG
seqs, stages, rng, n, L0, snp = stress.draw_case(seed, bool(os.environ.get("MANY")), os.environ.get("STAGES") == "3")
bf, orc = BlockFinder(seqs, device=0), Oracle(seqs)
win = None
if rng.random() < 0.3:
    win = int(rng.choice([1, 3, 64, 1000]))
    if not os.environ.get("NOWIN"):
        bf.set_window(win)
print("case", seed, "n", n, "L0", L0, "stages", stages, "snp", snp, "window", win, "NOWIN" if os.environ.get("NOWIN") else "")
for kk, dd in stages:
    a, b = bf.simplify_stage(kk, dd, 4), orc.simplify_stage(kk, dd, 4)
    (sa, pa), (sb, pb) = bf.state(), orc.state()
    st = bf.stats()
    print(" stage", kk, dd, "bulges gpu", a, "oracle", b, "sizes", [len(x) for x in sa] if hasattr(sa, "__len__") else sa, "equal", sa == sb,
          "rounds", st["rounds"], "replays", st["replays"], {k: st[k] for k in st if k in ("restarts", "dense", "chain_rounds", "parked", "iterations")})
    for i, (x, y) in enumerate(zip(pa, pb)):
        if not np.array_equal(x, y):
            x, y = np.asarray(x), np.asarray(y)
            if x.shape != y.shape:
                print("  array", i, "shapes", x.shape, y.shape)
            else:
                d = np.flatnonzero(x != y)
                print("  array", i, "differs at", len(d), "of", x.size, "first", d[:8].tolist())
if DBG:
    import ctypes as C
    L = C.CDLL(DBG)
    N = 1 << 20
    bufs = [(C.c_uint32 * N)() for _ in range(3)]
    L.sbl_dbg_idret(bufs[0], bufs[1], bufs[2], N, 1)
    log = (C.c_uint32 * (8 << 20))()
    L.sbl_dbg_log.restype = C.c_uint32
    nl = L.sbl_dbg_log(log, 1 << 20)
    np.save(os.environ.get("DBG_OUT", "gpurun_out/n/idret.npy").replace(".npy", "_log.npy"), np.ctypeslib.as_array(log)[:8 * nl].reshape(-1, 8).copy())
    np.save(os.environ.get("DBG_OUT", "gpurun_out/n/idret.npy"), np.stack([np.ctypeslib.as_array(b).copy() for b in bufs]))
bf.close()
