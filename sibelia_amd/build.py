"""Builds libsibelia_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsibelia_amd.so")
SOURCES = ["sbl_api.hip", "simplify.hip", "longk.hip", "shard.hip", "fasta_load.hip", "synteny.hip", "postprocess.hip"]
HEADERS = ["sbl_common.h", "sbl_ctx.h", "kmer_kernels.h", "kmer_bucket_kernels.h", "bulge_txn.h", "simplify_steps.h", "simplify_driver.h",
           os.path.join("..", "..", "include", "sibelia_amd.h")]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False) -> str:
    if force or stale():
        os.makedirs(LIBDIR, exist_ok=True)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB] + \
              [os.path.join(CSRC, s) for s in SOURCES]
        subprocess.run(cmd, check=True)
    return LIB
