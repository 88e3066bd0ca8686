"""ctypes binding of the CPU ORACLE (oracle/sibelia_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg --
never from the product package (sibelia_amd/).  `build()` compiles the C restatement with gcc.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsibelia_oracle.so")
SRC = os.path.join(HERE, "sibelia_oracle.c")

BLOCK_DTYPE = np.dtype([("id", "<i4"), ("chr", "<u4"), ("start", "<u8"), ("end", "<u8")])
INST_DTYPE = np.dtype([("id", "<u4"), ("chr", "<u4"), ("pos", "<u4")])
EDGE_DTYPE = np.dtype([("chr", "<u4"), ("strand", "<u4"), ("start_vertex", "<u4"), ("end_vertex", "<u4"),
                       ("pos", "<u4"), ("len", "<u4"), ("orig_pos", "<u4"), ("orig_len", "<u4"),
                       ("first_char", "S1"), ("_pad", "V3")])


SRC_CPP = os.path.join(HERE, "synteny_oracle.cpp")
SRC_CPP2 = os.path.join(HERE, "output_oracle.cpp")


def build(force: bool = False) -> str:
    srcs = [SRC, SRC_CPP, SRC_CPP2, os.path.join(HERE, "sibelia_oracle.h")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(x) for x in srcs):
        obj = os.path.join(HERE, "sibelia_oracle.o")
        subprocess.run(["gcc", "-O2", "-std=c99", "-fPIC", "-c", "-o", obj, SRC], check=True)
        subprocess.run(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-o", LIB, SRC_CPP, SRC_CPP2, obj], check=True)
        os.remove(obj)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_load.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
        L.orc_enumerate.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.orc_simplify_stage.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        L.orc_get_state.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.orc_nchr.argtypes = [C.c_void_p]
        L.orc_nchr.restype = C.c_uint32
        L.orc_list_edges.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.orc_generate_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.orc_postprocess.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_int,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.orc_serialize_graph.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.orc_kmer_hashes.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_force_long_k_path.argtypes = [C.c_void_p, C.c_int]
        L.orc_rand.argtypes = [C.c_void_p]
        L.orc_rand.restype = C.c_uint32
        L.orc_boost_order.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64)]
        L.orc_boost_order.restype = C.c_size_t
        L.orc_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _view(ptr, n, dtype):
    if not n:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Oracle:
    """Mirror of the reference's BlockFinder surface for the hot path, CPU oracle backed."""

    def __init__(self, seqs: Sequence[bytes]):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create())
        n = len(seqs)
        arr = (C.c_char_p * n)(*[bytes(s) for s in seqs])
        lens = (C.c_uint64 * n)(*[len(s) for s in seqs])
        self.L.orc_load(self.h, n, arr, lens)
        self._orig = [bytes(s) for s in seqs]           # originalChrList_ (GenerateSyntenyBlocks trims on the original sequences)

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def force_long_k_path(self, on: bool = True):
        self.L.orc_force_long_k_path(self.h, int(on))

    def rand(self) -> int:
        return self.L.orc_rand(self.h)

    def enumerate(self, k: int) -> Tuple[int, np.ndarray, np.ndarray]:
        bc = C.c_uint32()
        p, q = C.c_void_p(), C.c_void_p()
        n, m = C.c_uint64(), C.c_uint64()
        rc = self.L.orc_enumerate(self.h, k, C.byref(bc), C.byref(p), C.byref(n), C.byref(q), C.byref(m))
        if rc:
            raise ValueError("orc_enumerate failed: %d" % rc)
        return bc.value, _view(p.value, n.value, INST_DTYPE), _view(q.value, m.value, INST_DTYPE)

    def simplify_stage(self, k: int, min_branch: int, max_iter: int) -> int:
        b = C.c_uint64()
        rc = self.L.orc_simplify_stage(self.h, k, min_branch, max_iter, C.byref(b))
        if rc:
            raise ValueError("orc_simplify_stage failed: %d" % rc)
        return b.value

    def state(self) -> Tuple[List[bytes], List[np.ndarray]]:
        seqs, pos = [], []
        for c in range(self.L.orc_nchr(self.h)):
            s, p, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
            self.L.orc_get_state(self.h, c, C.byref(s), C.byref(p), C.byref(n))
            seqs.append(_view(s.value, n.value, np.dtype("u1")).tobytes())
            pos.append(_view(p.value, n.value, np.dtype("<u4")))
        return seqs, pos

    def list_edges(self, k: int) -> np.ndarray:
        e, n = C.c_void_p(), C.c_uint64()
        rc = self.L.orc_list_edges(self.h, k, C.byref(e), C.byref(n))
        if rc:
            raise ValueError("orc_list_edges failed: %d" % rc)
        return _view(e.value, n.value, EDGE_DTYPE)

    def generate_blocks(self, k: int, trim_k: int, min_size: int, shared_only: bool = False) -> np.ndarray:
        n = len(self._orig)
        arr = (C.c_char_p * n)(*self._orig)
        lens = (C.c_uint64 * n)(*[len(s) for s in self._orig])
        v, m = C.c_void_p(), C.c_uint64()
        rc = self.L.orc_generate_blocks(self.h, arr, lens, k, trim_k, min_size, int(shared_only), C.byref(v), C.byref(m))
        if rc:
            raise ValueError("orc_generate_blocks failed: %d" % rc)
        a = _view(v.value, m.value, BLOCK_DTYPE)
        self.L.orc_free(v)
        return a

    def postprocess(self, blocks: np.ndarray, names: Sequence[str], glue: bool = True):
        """GlueStripes + the three writers: (blocks, [blocks_coords.txt, genomes_permutations.txt, coverage_report.txt])."""
        b = np.ascontiguousarray(blocks, dtype=BLOCK_DTYPE)
        n = len(self._orig)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        sz = (C.c_uint64 * n)(*[len(s) for s in self._orig])
        ob, no = C.c_void_p(), C.c_uint64()
        tx, tl = (C.c_void_p * 3)(), (C.c_uint64 * 3)()
        rc = self.L.orc_postprocess(b.ctypes.data, len(b), n, nm, sz, int(glue), C.byref(ob), C.byref(no), tx, tl)
        if rc:
            raise ValueError("orc_postprocess failed: %d" % rc)
        out = _view(ob.value, no.value, BLOCK_DTYPE)
        texts = [C.string_at(tx[i], tl[i]) for i in range(3)]
        self.L.orc_free(ob)
        for i in range(3):
            self.L.orc_free(tx[i])
        return out, texts

    def serialize_graph(self, k: int) -> bytes:
        v, n = C.c_void_p(), C.c_uint64()
        rc = self.L.orc_serialize_graph(self.h, k, C.byref(v), C.byref(n))
        if rc:
            raise ValueError("orc_serialize_graph failed: %d" % rc)
        t = C.string_at(v, n.value)
        self.L.orc_free(v)
        return t

    def kmer_hashes(self, k: int) -> np.ndarray:
        v, n = C.c_void_p(), C.c_uint64()
        rc = self.L.orc_kmer_hashes(self.h, k, C.byref(v), C.byref(n))
        if rc:
            raise ValueError("orc_kmer_hashes failed: %d" % rc)
        a = _view(v.value, n.value, np.dtype("<u8"))
        self.L.orc_free(v)
        return a

    def last_timing(self) -> Tuple[float, float, float]:
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.L.orc_last_timing(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


def boost_order(keys: Sequence[int]) -> List[int]:
    L = lib()
    n = len(keys)
    a = (C.c_uint64 * n)(*keys)
    o = (C.c_uint64 * n)()
    m = L.orc_boost_order(a, n, o)
    return [o[i] for i in range(m)]
