"""ctypes driver of tests/hostsim/libhostsim.so (TEST INFRASTRUCTURE, see hostsim.hip)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhostsim.so")
SRC = os.path.join(HERE, "hostsim.hip")
CSRC = os.path.join(HERE, "..", "..", "sibelia_amd", "csrc")


def build(force: bool = False) -> str:
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("bulge_txn.h", "simplify_steps.h", "simplify_driver.h", "sbl_common.h")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(d) for d in deps):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.hostsim_free.argtypes = [C.c_void_p]
    return _lib


def stage(seqs: Sequence[bytes], opos: Sequence[np.ndarray], k: int, D: int, max_iter: int, bif_count: int,
          pos: np.ndarray, neg: np.ndarray, window: int = 64, order_mode: int = 0, arena_bytes: int = 1 << 16,
          slack: int = 1 << 16) -> Tuple[int, List[bytes], List[np.ndarray], dict]:
    L = lib()
    n = len(seqs)
    sarr = (C.c_char_p * n)(*[bytes(s) for s in seqs])
    ops = [np.ascontiguousarray(p, dtype=np.uint32) for p in opos]
    oarr = (C.c_void_p * n)(*[p.ctypes.data for p in ops])
    lens = (C.c_uint64 * n)(*[len(s) for s in seqs])
    p3 = np.ascontiguousarray(np.stack([pos["id"], pos["chr"], pos["pos"]], 1) if len(pos) else np.zeros((0, 3)), dtype=np.uint32)
    n3 = np.ascontiguousarray(np.stack([neg["id"], neg["chr"], neg["pos"]], 1) if len(neg) else np.zeros((0, 3)), dtype=np.uint32)
    oseq = (C.c_void_p * n)()
    oop = (C.c_void_p * n)()
    olen = (C.c_uint64 * n)()
    bulges = C.c_uint64()
    stats = (C.c_uint64 * 8)()
    rc = L.hostsim_stage(C.c_uint32(n), sarr, oarr, lens, C.c_uint32(k), C.c_uint32(D), C.c_uint32(max_iter), C.c_uint32(bif_count),
                         C.c_void_p(p3.ctypes.data), C.c_uint64(len(p3)), C.c_void_p(n3.ctypes.data), C.c_uint64(len(n3)),
                         C.c_uint32(window), C.c_int(order_mode), C.c_uint32(arena_bytes), C.c_uint32(slack),
                         oseq, oop, olen, C.byref(bulges), stats)
    if rc:
        raise RuntimeError("hostsim_stage failed: %d" % rc)
    rs, rp = [], []
    for i in range(n):
        m = olen[i]
        rs.append(C.string_at(oseq[i], m))
        rp.append(np.frombuffer(C.string_at(oop[i], 4 * m), dtype=np.uint32).copy())
        L.hostsim_free(oseq[i])
        L.hostsim_free(oop[i])
    return bulges.value, rs, rp, {"iterations": stats[0], "rounds": stats[1], "replays": stats[2], "solo": stats[3], "executed": stats[4], "grow_replays": stats[5]}
