"""Host-side mirror of the reference's BlockFinder surface over the C ABI (include/sibelia_amd.h).

`BlockFinder` keeps the reference's method names and argument meaning
(reference src/blockfinder.h:40-45): PerformGraphSimplifications(k, minBranchSize, maxIterations, f),
SerializeCondensedGraph(k, out), plus the enumeration / state accessors the parity tests need.
All compute happens in libsibelia_amd.so's HIP kernels; importing this module without the
built library, or constructing a BlockFinder without a GPU, fails loudly -- there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import formats
from .build import LIB

INST_DTYPE = np.dtype([("id", "<u4"), ("chr", "<u4"), ("pos", "<u4")])
EDGE_DTYPE = formats.EDGE_DTYPE
PROGRESS_FN = C.CFUNCTYPE(None, C.c_size_t, C.c_int, C.c_void_p)

EXPORTS = ["sbl_create", "sbl_destroy", "sbl_load", "sbl_enumerate", "sbl_simplify_stage", "sbl_get_state", "sbl_nchr",
           "sbl_list_edges", "sbl_last_stats", "sbl_last_error", "sbl_strerror", "sbl_set_window",
           "sbl_save_state", "sbl_restore_state", "sbl_load_fasta", "sbl_record_name", "sbl_kmer_hashes", "sbl_generate_blocks", "sbl_postprocess", "sbl_serialize_graph",
           "sbl_set_tempfile_mode", "sbl_rand_advance", "sbl_shard_layout", "sbl_shard_exchange_plan", "sbl_glue_stripes", "sbl_comm_unique_id", "sbl_comm_attach_rccl", "sbl_comm_attach_local", "sbl_comm_detach",
           "sbl_longk_slices", "sbl_longk_value_bounds", "sbl_longk_owner", "sbl_longk_halo_plan"]


class StageStats(C.Structure):
    _fields_ = [("strand_kmers", C.c_uint64), ("bif_count", C.c_uint64), ("instances", C.c_uint64), ("bulges", C.c_uint64),
                ("iterations", C.c_uint32), ("rounds", C.c_uint32), ("replays", C.c_uint32), ("grow_replays", C.c_uint32),
                ("enumerate_ms", C.c_double), ("simplify_ms", C.c_double), ("copyback_ms", C.c_double), ("total_ms", C.c_double),
                ("kmer_table_ms", C.c_double), ("kmer_table_bytes", C.c_uint64),
                ("snapshot_ms", C.c_double), ("reserve_ms", C.c_double), ("commit_ms", C.c_double), ("probe_ms", C.c_double),
                ("executed", C.c_uint64), ("transactions", C.c_uint64),
                ("exchange_ms", C.c_double), ("exchange_bytes", C.c_uint64), ("chain_transactions", C.c_uint64),
                ("commit_event_ms", C.c_double), ("commit_event_launches", C.c_uint64),
                ("dict_checked", C.c_uint64), ("dict_mismatches", C.c_uint64),
                ("ro_ranks", C.c_uint64), ("verdict_ms", C.c_double), ("verdict_bytes", C.c_uint64),
                ("longk_path", C.c_uint64), ("fp_verified", C.c_uint64), ("device_bytes", C.c_uint64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class SibeliaError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen libsibelia_amd.so (built in-tree by sibelia_amd.build / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise SibeliaError("libsibelia_amd.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is mandatory, there is no host fallback)")
        L = C.CDLL(LIB)
        L.sbl_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.sbl_destroy.argtypes = [C.c_void_p]
        L.sbl_destroy.restype = None
        L.sbl_load.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
        L.sbl_enumerate.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.sbl_simplify_stage.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.sbl_get_state.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.sbl_nchr.argtypes = [C.c_void_p]
        L.sbl_nchr.restype = C.c_uint32
        L.sbl_list_edges.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.sbl_last_stats.argtypes = [C.c_void_p, C.POINTER(StageStats)]
        L.sbl_last_error.argtypes = [C.c_void_p]
        L.sbl_last_error.restype = C.c_char_p
        L.sbl_strerror.argtypes = [C.c_int]
        L.sbl_strerror.restype = C.c_char_p
        L.sbl_set_window.argtypes = [C.c_void_p, C.c_uint32]
        L.sbl_save_state.argtypes = [C.c_void_p]
        L.sbl_restore_state.argtypes = [C.c_void_p]
        L.sbl_load_fasta.argtypes = [C.c_void_p, C.c_char_p]
        L.sbl_record_name.argtypes = [C.c_void_p, C.c_uint32]
        L.sbl_record_name.restype = C.c_char_p
        L.sbl_generate_blocks.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.sbl_postprocess.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        L.sbl_serialize_graph.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.sbl_kmer_hashes.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.sbl_comm_unique_id.argtypes = [C.c_void_p]
        L.sbl_comm_attach_rccl.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.sbl_group_create_local.argtypes = [C.c_uint32]
        L.sbl_group_create_local.restype = C.c_void_p
        L.sbl_group_destroy.argtypes = [C.c_void_p]
        L.sbl_group_destroy.restype = None
        L.sbl_comm_attach_local.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.sbl_comm_detach.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _view(ptr, n, dtype):
    if not n:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class BlockFinder:
    """SyntenyFinder::BlockFinder for the hot path, backed by one MI355X.

    seqs: upper-case sequences as delivered by the reference FASTA reader (one per FASTARecord)."""

    def __init__(self, seqs: Sequence[bytes], device: int = -1, fasta: Optional[str] = None):
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.sbl_create(C.byref(self.h), device)
        if rc:
            raise SibeliaError("sbl_create: " + self.L.sbl_strerror(rc).decode())
        if fasta is not None:          # FASTAReader + Init on the device (reference src/fasta.cpp:23-104, src/blockfinder.cpp:65-76)
            self._check(self.L.sbl_load_fasta(self.h, os.fsencode(fasta)), "sbl_load_fasta")
            return
        n = len(seqs)
        arr = (C.c_char_p * n)(*[bytes(s) for s in seqs])
        lens = (C.c_uint64 * n)(*[len(s) for s in seqs])
        self._check(self.L.sbl_load(self.h, n, arr, lens), "sbl_load")

    @classmethod
    def from_fasta(cls, path: str, device: int = -1) -> "BlockFinder":
        return cls((), device=device, fasta=path)

    def record_names(self) -> List[str]:
        return [self.L.sbl_record_name(self.h, i).decode() for i in range(self.L.sbl_nchr(self.h))]

    def _check(self, rc, what):
        if rc:
            msg = self.L.sbl_last_error(self.h).decode() if self.h else ""
            raise SibeliaError("%s: %s (%s)" % (what, self.L.sbl_strerror(rc).decode(), msg))

    def close(self):
        if getattr(self, "h", None):
            self.L.sbl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference surface ---------------------------------------------------------------
    def PerformGraphSimplifications(self, k: int, minBranchSize: int, maxIterations: int,
                                    f: Optional[Callable[[int, int], None]] = None) -> int:
        b = C.c_uint64()
        cb = PROGRESS_FN(lambda p, s, u: f(p, s)) if f else None
        rc = self.L.sbl_simplify_stage(self.h, k, minBranchSize, maxIterations, C.cast(cb, C.c_void_p) if cb else None, None, C.byref(b))
        self._check(rc, "sbl_simplify_stage")
        return b.value

    def SerializeCondensedGraph(self, k: int, out) -> None:
        out.write(formats.dot_text(self.list_edges(k)).decode("latin1"))

    # ---- backend protocol shared with the oracle wrapper (tests/vectors.py) ------------------
    def enumerate(self, k: int) -> Tuple[int, np.ndarray, np.ndarray]:
        bc = C.c_uint32()
        p, q = C.c_void_p(), C.c_void_p()
        n, m = C.c_uint64(), C.c_uint64()
        self._check(self.L.sbl_enumerate(self.h, k, C.byref(bc), C.byref(p), C.byref(n), C.byref(q), C.byref(m)), "sbl_enumerate")
        return bc.value, _view(p.value, n.value, INST_DTYPE), _view(q.value, m.value, INST_DTYPE)

    def simplify_stage(self, k: int, min_branch: int, max_iter: int) -> int:
        return self.PerformGraphSimplifications(k, min_branch, max_iter)

    def state(self) -> Tuple[List[bytes], List[np.ndarray]]:
        seqs, pos = [], []
        for c in range(self.L.sbl_nchr(self.h)):
            s, p, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
            self._check(self.L.sbl_get_state(self.h, c, C.byref(s), C.byref(p), C.byref(n)), "sbl_get_state")
            seqs.append(_view(s.value, n.value, np.dtype("u1")).tobytes())
            pos.append(_view(p.value, n.value, np.dtype("<u4")))
        return seqs, pos

    def state_views(self) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """The same state as zero-copy numpy views of the library's pinned staging buffer (one bulk device-to-host copy of
        ch[] + op[], 5 B per base; chromosomes are slices of it).  Borrowed: valid until the next mutating call."""
        seqs, pos = [], []
        for c in range(self.L.sbl_nchr(self.h)):
            s, p, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
            self._check(self.L.sbl_get_state(self.h, c, C.byref(s), C.byref(p), C.byref(n)), "sbl_get_state")
            if not n.value:
                seqs.append(np.zeros(0, np.uint8)); pos.append(np.zeros(0, "<u4")); continue
            seqs.append(np.frombuffer((C.c_char * n.value).from_address(s.value), dtype=np.uint8, count=n.value))
            pos.append(np.frombuffer((C.c_char * (4 * n.value)).from_address(p.value), dtype="<u4", count=n.value))
        return seqs, pos

    def list_edges(self, k: int) -> np.ndarray:
        e, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.sbl_list_edges(self.h, k, C.byref(e), C.byref(n)), "sbl_list_edges")
        return _view(e.value, n.value, EDGE_DTYPE)

    def GenerateSyntenyBlocks(self, k: int, trimK: int, minSize: int, sharedOnly: bool = False) -> np.ndarray:
        """BlockFinder::GenerateSyntenyBlocks (reference src/blockfinder.h:43): BlockInstance records (id, chr, start, end)."""
        b, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.sbl_generate_blocks(self.h, k, trimK, minSize, int(sharedOnly), C.byref(b), C.byref(n)), "sbl_generate_blocks")
        return _view(b.value, n.value, formats.BLOCK_DTYPE)

    generate_blocks = GenerateSyntenyBlocks

    def postprocess(self, names: Optional[Sequence[str]] = None, glue: bool = True):
        """GlueStripes (reference src/postprocessor.cpp:37-154) on the blocks of the last GenerateSyntenyBlocks + the texts of
        blocks_coords.txt, genomes_permutations.txt, coverage_report.txt (src/outputgenerator.cpp:162-233)."""
        nm = None
        if names is not None:
            if len(names) != self.L.sbl_nchr(self.h):      # the C entry point reads one name per loaded record (include/sibelia_amd.h)
                raise ValueError("postprocess: %d names for %d records" % (len(names), self.L.sbl_nchr(self.h)))
            nm = (C.c_char_p * len(names))(*[x.encode() for x in names])
        b, n = C.c_void_p(), C.c_uint64()
        t = [C.c_char_p() for _ in range(3)]
        self._check(self.L.sbl_postprocess(self.h, int(glue), nm, C.byref(b), C.byref(n), C.byref(t[0]), C.byref(t[1]), C.byref(t[2])), "sbl_postprocess")
        return _view(b.value, n.value, formats.BLOCK_DTYPE), [x.value for x in t]

    def serialize_graph(self, k: int) -> bytes:
        """BlockFinder::SerializeGraph (reference src/blockfinder.h:41): DOT text of the uncondensed graph."""
        t, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.sbl_serialize_graph(self.h, k, C.byref(t), C.byref(n)), "sbl_serialize_graph")
        return C.string_at(t, n.value)

    def kmer_hashes(self, k: int) -> np.ndarray:
        """H0: hashes of the reference's hashing.h for every k-mer, strand 0 then 1, chromosomes ascending, walk order."""
        v, n = C.c_void_p(), C.c_uint64()
        self._check(self.L.sbl_kmer_hashes(self.h, k, C.byref(v), C.byref(n)), "sbl_kmer_hashes")
        return _view(v.value, n.value, np.dtype("<u8"))

    def stats(self) -> dict:
        s = StageStats()
        self.L.sbl_last_stats(self.h, C.byref(s))
        return s.as_dict()

    def save_state(self) -> None:
        self._check(self.L.sbl_save_state(self.h), "sbl_save_state")

    def restore_state(self) -> None:
        self._check(self.L.sbl_restore_state(self.h), "sbl_restore_state")

    def set_tempfile_mode(self, on: bool = True) -> None:
        """BlockFinder(chrList, tempDir) of the reference: keep the rand() stream in step with its temp-file names (include/sibelia_amd.h)."""
        self.L.sbl_set_tempfile_mode.argtypes = [C.c_void_p, C.c_int]
        self._check(self.L.sbl_set_tempfile_mode(self.h, int(on)), "sbl_set_tempfile_mode")

    def rand_advance(self, n: int) -> None:
        self.L.sbl_rand_advance.argtypes = [C.c_void_p, C.c_uint64]
        self._check(self.L.sbl_rand_advance(self.h, int(n)), "sbl_rand_advance")

    def set_window(self, w: int) -> None:
        self.L.sbl_set_window(self.h, w)

    # ---- multi-GPU: hash-prefix sharded enumeration (include/sibelia_amd.h, csrc/shard.hip) ----
    def attach_rccl(self, rank: int, nranks: int, unique_id: bytes) -> None:
        """Collective over all ranks; unique_id = comm_unique_id() of rank 0, distributed by the host."""
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        self._check(self.L.sbl_comm_attach_rccl(self.h, rank, nranks, buf), "sbl_comm_attach_rccl")

    def attach_local(self, group: "LocalGroup", rank: int) -> None:
        self._group = group                         # keep the group alive as long as the context uses it
        self._check(self.L.sbl_comm_attach_local(self.h, group.h, rank), "sbl_comm_attach_local")

    def detach(self) -> None:
        self._check(self.L.sbl_comm_detach(self.h), "sbl_comm_detach")
        self._group = None


COMM_ID_BYTES = 128


def shard_layout(nranks: int, rank: int, bits: int, ntiles: int) -> Tuple[np.ndarray, Tuple[int, int]]:
    """Device-free layout arithmetic of the sharded k-mer table (csrc/shard.hip uses the same entry point): first bucket of
    every owner (nranks + 1 values; owner(b) = (b * nranks) >> bits) and the tile range this rank scans."""
    L = load_library()
    fb = (C.c_uint32 * (nranks + 1))()
    tr = (C.c_uint64 * 2)()
    L.sbl_shard_layout.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
    rc = L.sbl_shard_layout(nranks, rank, bits, ntiles, fb, tr)
    if rc:
        raise SibeliaError("sbl_shard_layout: " + L.sbl_strerror(rc).decode())
    return np.array(fb, dtype=np.uint32), (int(tr[0]), int(tr[1]))


def shard_exchange_plan(nranks: int, rank: int, count: np.ndarray, send_at: np.ndarray, record_bytes: int = 8):
    """Byte counts / offsets of the one all-to-all: count[p, q] = records rank p holds for owner q (all-gathered), send_at = where
    the owners' ranges start in this rank's partitioned arrays.  Returns (sbytes, soff, rbytes, roff, nrecv)."""
    L = load_library()
    cnt = np.ascontiguousarray(count, dtype=np.uint64).reshape(nranks, nranks)
    sa = np.ascontiguousarray(send_at, dtype=np.uint32)
    out = [np.zeros(nranks, dtype=np.uint64) for _ in range(4)]
    nrecv = C.c_uint64()
    L.sbl_shard_exchange_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64] + [C.c_void_p] * 4 + [C.c_void_p]
    rc = L.sbl_shard_exchange_plan(nranks, rank, cnt.ctypes.data, sa.ctypes.data, record_bytes, *[o.ctypes.data for o in out], C.byref(nrecv))
    if rc:
        raise SibeliaError("sbl_shard_exchange_plan: " + L.sbl_strerror(rc).decode())
    return out[0], out[1], out[2], out[3], int(nrecv.value)


def longk_slices(nranks: int, np_: int) -> np.ndarray:
    """Position slices of the sharded rank doubling (csrc/longk.hip): first[r] = np * r / nranks, nranks + 1 values."""
    L = load_library()
    out = np.zeros(nranks + 1, dtype=np.uint64)
    L.sbl_longk_slices.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p]
    if L.sbl_longk_slices(nranks, np_, out.ctypes.data):
        raise SibeliaError("sbl_longk_slices")
    return out


def longk_value_bounds(nranks: int, maxvalue: int) -> np.ndarray:
    L = load_library()
    out = np.zeros(nranks + 1, dtype=np.uint64)
    L.sbl_longk_value_bounds.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p]
    if L.sbl_longk_value_bounds(nranks, maxvalue, out.ctypes.data):
        raise SibeliaError("sbl_longk_value_bounds")
    return out


def longk_owner(bounds: np.ndarray, x: int) -> int:
    L = load_library()
    b = np.ascontiguousarray(bounds, dtype=np.uint64)
    o = C.c_uint32()
    L.sbl_longk_owner.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    if L.sbl_longk_owner(len(b) - 1, b.ctypes.data, int(x), C.byref(o)):
        raise SibeliaError("sbl_longk_owner")
    return int(o.value)


def longk_halo_plan(nranks: int, rank: int, np_: int, H: int):
    """(sbytes, soff, rbytes, roff) of the halo fetch of the sharded rank doubling: 4-B ranks, offsets into the sender's slice /
    the receiver's halo."""
    L = load_library()
    out = [np.zeros(nranks, dtype=np.uint64) for _ in range(4)]
    L.sbl_longk_halo_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64] + [C.c_void_p] * 4
    if L.sbl_longk_halo_plan(nranks, rank, np_, H, *[o.ctypes.data for o in out]):
        raise SibeliaError("sbl_longk_halo_plan")
    return out


def glue_stripes(blocks: np.ndarray, nchr: int) -> np.ndarray:
    """Postprocessor::GlueStripes (reference src/postprocessor.cpp:37-154) on a block array (formats.BLOCK_DTYPE); host bookkeeping only."""
    L = load_library()
    b = np.ascontiguousarray(blocks, dtype=formats.BLOCK_DTYPE).copy()
    n = C.c_uint64(len(b))
    L.sbl_glue_stripes.restype = C.c_int
    L.sbl_glue_stripes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32]
    rc = L.sbl_glue_stripes(b.ctypes.data if len(b) else None, C.byref(n), int(nchr))
    if rc:
        raise SibeliaError("sbl_glue_stripes failed: %d" % rc)
    return b[:n.value]


def comm_unique_id() -> bytes:
    L = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = L.sbl_comm_unique_id(buf)
    if rc:
        raise SibeliaError("sbl_comm_unique_id: " + L.sbl_strerror(rc).decode())
    return buf.raw


class LocalGroup:
    """Contexts of one process, one host thread per virtual rank (tests: several ranks on one GPU)."""

    def __init__(self, nranks: int):
        self.L = load_library()
        self.n = nranks
        self.h = C.c_void_p(self.L.sbl_group_create_local(nranks))
        if not self.h:
            raise SibeliaError("sbl_group_create_local failed")

    def run(self, fns):
        """Run one callable per rank concurrently (the calls are collective) and return their results."""
        import threading
        out, err = [None] * self.n, [None] * self.n

        def body(i):
            try:
                out[i] = fns[i]()
            except BaseException as e:          # noqa: BLE001 - reported to the caller below
                err[i] = e
        th = [threading.Thread(target=body, args=(i,)) for i in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in err:
            if e is not None:
                raise e
        return out

    def __del__(self):
        try:
            if self.h:
                self.L.sbl_group_destroy(self.h)
                self.h = None
        except Exception:
            pass
