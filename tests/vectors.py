"""Golden-vector replay shared by the oracle tests (CPU) and the HIP parity tests (GPU).

A backend is any object with the reference-shaped surface
    enumerate(k) -> (bif_count, pos[n] {id,chr,pos}, neg[m] {id,chr,pos})
    simplify_stage(k, min_branch, max_iter) -> bulges
    state() -> (list[bytes], list[np.ndarray uint32])
    list_edges(k) -> structured edge array
    generate_blocks(k, trim_k, min_size, shared_only) -> structured block array (N2, reference src/synteny.cpp)
    kmer_hashes(k) -> flat uint64 array (H0, reference src/hashing.h)
"""
from __future__ import annotations

import json
import os
from typing import Callable, List, Sequence

import numpy as np

from sibelia_amd import formats as F
from sibelia_amd import workloads as W

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def load_vectors() -> List[dict]:
    with open(os.path.join(GOLDEN, "vectors.json")) as f:
        return json.load(f)["vectors"]


def vector_input(v: dict) -> List[bytes]:
    src = v["input"]
    if src["kind"] == "literal":
        seqs = [s.encode() for s in src["seqs"]]
    elif src["kind"] == "small_case":
        seqs = W.small_case(src["seed"])[0]
    elif src["kind"] == "cascade_case":
        seqs = W.cascade_case(src["seed"])[0]
    elif src["kind"] == "fasta":
        seqs = W.read_fasta(os.path.join(GOLDEN, "data", src["file"]))[1]
    elif src["kind"] == "gen_strains":
        seqs = W.gen_strains(**src["args"])
    else:
        raise ValueError(src["kind"])
    assert W.input_digest(seqs) == v["input_sha256"], "input generator drifted for " + v["name"]
    return seqs


def _inst_cols(a: np.ndarray) -> np.ndarray:
    return np.stack([a["id"], a["chr"], a["pos"]], axis=1).astype("<u4") if len(a) else np.zeros((0, 3), "<u4")


def run_cmd(backend, cmd: str) -> bytes:
    p = cmd.split(":")
    if p[0] == "enum":
        bc, pos, neg = backend.enumerate(int(p[1]))
        return F.enum_bytes(bc, _inst_cols(pos), _inst_cols(neg))
    if p[0] == "stage":
        bulges = backend.simplify_stage(int(p[1]), int(p[2]), int(p[3]))
        seqs, opos = backend.state()
        return F.state_bytes(bulges, seqs, opos)
    if p[0] == "blocks":
        return F.blocks_bytes(backend.generate_blocks(int(p[1]), int(p[2]), int(p[3]), bool(int(p[4]))))
    if p[0] == "write":                                  # N4: blocks after GlueStripes + the three report texts
        k, tk, ms, sh, gl = (int(x) for x in p[1:6])
        blocks = backend.generate_blocks(k, tk, ms, bool(sh))
        names = ["seq%d" % i for i in range(len(backend.state()[0]))]      # W.write_fasta's descriptions (what the reference read)
        out, texts = backend.postprocess(blocks, names, bool(gl)) if hasattr(backend, "_orig") else backend.postprocess(names, bool(gl))
        return F.blocks_bytes(out) + b"".join(__import__("struct").pack("<Q", len(t)) + t for t in texts)
    if p[0] == "graph":
        return backend.serialize_graph(int(p[1]))
    if p[0] == "hash":
        k = int(p[1])
        return F.hash_bytes(backend.kmer_hashes(k), [len(x) for x in backend.state()[0]], k)
    if p[0] == "dot":
        return F.dot_text(backend.list_edges(int(p[1])))
    raise ValueError(cmd)


def replay(v: dict, make_backend: Callable[[Sequence[bytes]], object]) -> None:
    seqs = vector_input(v)
    b = make_backend(seqs)
    try:
        for i, o in enumerate(v["outputs"]):
            got = run_cmd(b, o["cmd"])
            assert len(got) == o["size"] and F.sha256(got) == o["sha256"], \
                "%s: output %d (%s) differs from the reference" % (v["name"], i, o["cmd"])
    finally:
        if hasattr(b, "close"):
            b.close()
