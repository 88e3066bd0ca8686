"""The product's bulge-removal transaction code (sibelia_amd/csrc/bulge_txn.h + the ordered-commit
driver) executed on the host one thread at a time (tests/hostsim) and compared with the oracle.

Covers what the GPU parity tests cannot show without hardware: Boost-order groups, lazy erase,
double interpolation, reservation/validation/replay logic, and independence of the result from the
window size and from the execution order inside a round."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from sibelia_amd import workloads as W
from tests.hostsim import hostsim as H


def _check(seqs, k, D, window, order, arena=1 << 16, it=4):
    o = Oracle(seqs)
    bc, p, n = o.enumerate(k)
    o2 = Oracle(seqs)
    b = o2.simplify_stage(k, D, it)
    es, ep = o2.state()
    op = [np.arange(len(s), dtype=np.uint32) for s in seqs]
    hb, hs, hp, st = H.stage(seqs, op, k, D, it, bc, p, n, window=window, order_mode=order, arena_bytes=arena)
    assert hb == b and hs == es and all(np.array_equal(a, c) for a, c in zip(hp, ep)), st
    return st


def _seeds(limit=14):
    # seeds of the golden small cases (the reference finished them quickly), pure ACGT, k <= 32
    import json, os
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))["vectors"]
    out = []
    for v in vec:
        if v["input"]["kind"] != "small_case":
            continue
        seed = v["input"]["seed"]
        seqs, k, D = W.small_case(seed)
        bulges = [o.get("bulges", 0) for o in v["outputs"] if o["cmd"].startswith("stage")]
        if k <= 32 and all(set(x) <= set(b"ACGT") for x in seqs) and 0 < bulges[0] < 3000:
            out.append(seed)
        if len(out) >= limit:
            break
    return out


SEEDS = _seeds()


@pytest.mark.parametrize("seed", SEEDS)
def test_transactions_match_oracle_small(seed):
    seqs, k, D = W.small_case(seed)
    _check(seqs, k, D, window=1, order=0)
    _check(seqs, k, D, window=37, order=2)


def test_transactions_match_oracle_strains():
    seqs = W.gen_strains(L0=30_000, n=4, seed=7, inv_min=1000, inv_max=4000)
    st = _check(seqs, 25, 150, window=4096, order=2)
    assert st["executed"] > 0
    _check(seqs, 25, 150, window=64, order=1)


def test_lazy_boost_map_gives_the_same_order(monkeypatch):
    """AnyBulges' unordered_map restated lazily (bulge_txn.h: ABuild::lazy, what the kernels' wave_any_bulges does): the insertions
    are only logged and go through the Boost restatement when a call has two or more bulge groups -- cases with many such calls
    (small alphabets, low k) must come out exactly as with the eager map"""
    monkeypatch.setenv("HOSTSIM_LAZY_MAP", "1")
    for seed in SEEDS[:6]:
        seqs, k, D = W.small_case(seed)
        _check(seqs, k, D, window=37, order=2)
    seqs = W.gen_strains(L0=20_000, n=6, seed=17, snp=0.03, inv_min=500, inv_max=2000)
    _check(seqs, 16, 120, window=512, order=1)


def test_small_arena_goes_through_the_big_path():
    seqs, k, D = W.small_case(2)            # k=4: ids with many instances overflow a tiny arena
    st = _check(seqs, k, D, window=16, order=0, arena=1 << 12)
    assert st["solo"] > 0


# ---- the densest golden vectors (>= 8000 collapses at k = 3 .. 10 on a few hundred bases: ids with thousands of instances,
# every transaction in conflict with every other): the regime where the ordering logic is most fragile.  All their stage
# commands run through the product's transaction code here, against the reference's fixtures AND the oracle.
def _huge():
    import json, os
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))["vectors"]
    return [v for v in vec if v["name"].startswith("small/") and sum(o.get("bulges", 0) for o in v["outputs"]) >= 8000]


HUGE = _huge()
# (small/078's second stage -- k = 3, D = 144 on 1.4 kbp: ids grow to ~8000 instances and nearly every cached window sees every
# collapse -- took ~9 min while every such window was rescanned after every collapse; large ids now rescan windows on demand.)
SLOW_STAGES = set()


@pytest.mark.parametrize("v", HUGE, ids=[v["name"] for v in HUGE])
def test_transactions_match_reference_on_the_densest_vectors(v):
    from tests import vectors as V
    from sibelia_amd import formats as F
    ACGT = set(b"ACGT")
    seqs = V.vector_input(v)
    o, rng = Oracle(seqs), Oracle([])                      # rng: a second glibc rand() stream kept in step with the oracle's
    cur = [bytes(s) for s in seqs]
    op = [np.arange(len(s), dtype=np.uint32) for s in seqs]
    for out in v["outputs"]:
        p = out["cmd"].split(":")
        namb = sum(1 for s in cur for c in s if c not in ACGT)
        if p[0] != "stage":                                 # a fresh IndexedSequence sanitises a COPY: rand() advances, the state does not
            for _ in range(namb):
                rng.rand()
            (o.enumerate if p[0] == "enum" else o.list_edges)(int(p[1]))
            continue
        if (v["name"], out["cmd"]) in SLOW_STAGES:
            break
        k, D, it = int(p[1]), int(p[2]), int(p[3])
        if namb:                                            # reference src/indexedsequence.cpp:31-37
            cur = [bytes((c if c in ACGT else b"ACGT"[rng.rand() % 4]) for c in s) for s in cur]
        bc, pp, nn = Oracle(cur).enumerate(k)
        hb, hs, hp, st = H.stage(cur, op, k, D, it, bc, pp, nn, window=64, order_mode=2, arena_bytes=1 << 16, slack=1 << 21)
        ob = o.simplify_stage(k, D, it)
        es, ep = o.state()
        assert hb == ob == out["bulges"] and hs == es and all(np.array_equal(a, c) for a, c in zip(hp, ep)), (out["cmd"], st)
        got = F.state_bytes(hb, hs, hp)
        assert len(got) == out["size"] and F.sha256(got) == out["sha256"], "differs from the reference fixture"
        cur, op = hs, hp
