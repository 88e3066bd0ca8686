"""Canonical byte serialisations of hot-path results (shared by tests, bench and smoke).

These are the exact layouts the golden-vector generator writes (tests/golden/gen/ref_dump.cpp),
so that sha256(serialise(result)) can be compared with tests/golden/vectors.json:

  enum   u32 bif_count | for strand in (+,-): u64 n, n x (u32 id, u32 chr, u32 pos) in (chr,pos) order
  state  u64 bulges | u32 nchr | per chr: u64 len, len bytes, len x u32 original positions
  dot    text of BlockFinder::SerializeCondensedGraph (reference src/serialization.cpp:88-110)
  blocks u64 n | n x (i32 signed block id, u32 chr, u64 start, u64 end): BlockFinder::GenerateSyntenyBlocks' result, in its order
  write  blocks (as above, after GlueStripes) | 3 x (u64 len, text): blocks_coords.txt, genomes_permutations.txt, coverage_report.txt
  hash   for strand in (+,-), per chromosome: u64 n, n x u64 k-mer hashes of the reference's hashing.h in walk order
"""
from __future__ import annotations

import hashlib
import struct
from typing import Sequence

import numpy as np


def enum_bytes(bif_count: int, pos: np.ndarray, neg: np.ndarray) -> bytes:
    """pos/neg: uint32 arrays of shape (n, 3) with columns (id, chr, pos)."""
    out = [struct.pack("<I", bif_count)]
    for a in (pos, neg):
        a = np.ascontiguousarray(a, dtype="<u4").reshape(-1, 3)
        out.append(struct.pack("<Q", a.shape[0]))
        out.append(a.tobytes())
    return b"".join(out)


def state_bytes(bulges: int, seqs: Sequence[bytes], opos: Sequence[np.ndarray]) -> bytes:
    out = [struct.pack("<QI", bulges, len(seqs))]
    for s, p in zip(seqs, opos):
        out.append(struct.pack("<Q", len(s)))
        out.append(bytes(s))
        out.append(np.ascontiguousarray(p, dtype="<u4").tobytes())
    return b"".join(out)


BLOCK_DTYPE = np.dtype([("id", "<i4"), ("chr", "<u4"), ("start", "<u8"), ("end", "<u8")])


def blocks_bytes(blocks: np.ndarray) -> bytes:
    b = np.ascontiguousarray(blocks, dtype=BLOCK_DTYPE)
    return struct.pack("<Q", len(b)) + b.tobytes()


def hash_bytes(values: np.ndarray, lens: Sequence[int], k: int) -> bytes:
    """values: flat uint64 array, strand 0 then strand 1, chromosomes ascending; lens: chromosome lengths."""
    values = np.ascontiguousarray(values, dtype="<u8")
    out, at = [], 0
    for _ in range(2):
        for n in lens:
            m = n - k + 1 if n >= k else 0
            out.append(struct.pack("<Q", m))
            out.append(values[at:at + m].tobytes())
            at += m
    assert at == len(values)
    return b"".join(out)


EDGE_DTYPE = np.dtype([("chr", "<u4"), ("strand", "<u4"), ("start_vertex", "<u4"), ("end_vertex", "<u4"),
                       ("pos", "<u4"), ("len", "<u4"), ("orig_pos", "<u4"), ("orig_len", "<u4"),
                       ("first_char", "S1"), ("_pad", "V3")])


def dot_text(edges: np.ndarray) -> bytes:
    """Edge records (EDGE_DTYPE) -> the reference's DOT text (src/serialization.cpp:92-109)."""
    lines = ["digraph G", "{", "rankdir=LR"]
    for e in edges:
        lines.append('%d -> %d [color="%s", label="chr=%d pos=%d len=%d orpos=%d orlen=%d  ch=\'%s\'"];' % (
            e["start_vertex"], e["end_vertex"], "blue" if e["strand"] == 0 else "red",
            np.int32(e["chr"]), np.int32(e["pos"]), np.int32(e["len"]), np.int32(e["orig_pos"]), np.int32(e["orig_len"]),
            e["first_char"].decode("latin1")))
    lines.append("}")
    return ("\n".join(lines) + "\n").encode("latin1")


def sha256(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()
