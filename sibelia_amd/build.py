"""Builds libsibelia_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

One object per translation unit (sibelia_amd/lib/obj/*.o, compiled in parallel, recompiled when the source or any header is newer),
then one link: touching longk.hip does not recompile the round kernels, and the four kernel units of the simplification
(graphbuild / snapshot / rounds / commit .hip, split out of simplify.hip in round 5) compile side by side."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libsibelia_amd.so")
SOURCES = ["sbl_api.hip", "simplify.hip", "graphbuild.hip", "snapshot.hip", "rounds.hip", "commit.hip", "longk.hip", "longk_fp.hip", "shard.hip", "fasta_load.hip", "synteny.hip", "postprocess.hip"]
HEADERS = ["sbl_common.h", "sbl_ctx.h", "sbl_comm.h", "kmer_kernels.h", "kmer_bucket_kernels.h", "bulge_txn.h", "simplify_steps.h", "simplify_driver.h",
           "simplify_device.h", "simplify_walks.h", "simplify_kernels.h",
           os.path.join("..", "..", "include", "sibelia_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _sources():
    missing = [s for s in SOURCES if not os.path.exists(os.path.join(CSRC, s))]
    if missing:      # a renamed or deleted translation unit must be a build error, not a library with undefined symbols
        raise FileNotFoundError("sibelia_amd/build.py lists sources that do not exist: " + ", ".join(missing))
    return list(SOURCES)


def _obj(src: str) -> str:
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")


def _newest_header() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in HEADERS)


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in _sources() + HEADERS)


def build(force: bool = False) -> str:
    if force or stale():
        os.makedirs(OBJDIR, exist_ok=True)
        hdr = _newest_header()
        todo = []
        for s in _sources():
            o = _obj(s)
            if force or not os.path.exists(o) or os.path.getmtime(o) < max(hdr, os.path.getmtime(os.path.join(CSRC, s))):
                todo.append(s)

        def cc(s):
            subprocess.run(["hipcc"] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", _obj(s)], check=True)

        with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as ex:
            list(ex.map(cc, todo))
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in _sources()], check=True)
    return LIB
