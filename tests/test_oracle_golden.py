"""Pins the CPU oracle (oracle/sibelia_oracle.c) to outputs of the unmodified reference.

Every vector in tests/golden/vectors.json was produced by the reference binary
(tests/golden/gen/): bifurcation ids + instances, post-stage sequences + original
positions + bulge counts, and condensed-graph DOT text must match bit for bit.
"""
import pytest

from oracle.oracle import Oracle, boost_order
from tests import vectors as V

VECS = V.load_vectors()
FAST = [v for v in VECS if not v["name"].startswith(("real/", "synth/strains2", "synth/strains8"))]
SLOW = [v for v in VECS if v["name"].startswith(("real/", "synth/strains2"))]


@pytest.mark.parametrize("v", FAST, ids=[v["name"] for v in FAST])
def test_oracle_matches_reference(v):
    V.replay(v, Oracle)


@pytest.mark.slow
@pytest.mark.parametrize("v", SLOW, ids=[v["name"] for v in SLOW])
def test_oracle_matches_reference_genomes(v):
    V.replay(v, Oracle)


def test_glibc_rand_known_answers():
    # first values of unseeded glibc rand()
    o = Oracle([b"ACGT"])
    assert [o.rand() for _ in range(3)] == [1804289383, 846930886, 1681692777]


def test_long_k_grouping_equals_packed_code_grouping():
    # rank-doubling path (k > 32) cross-checked against the 2-bit code path at small k
    from sibelia_amd import workloads as W
    for seed in (1, 2, 5, 8, 13):
        seqs, k, _ = W.small_case(seed)
        if k > 32:
            continue
        a, b = Oracle(seqs), Oracle(seqs)
        b.force_long_k_path()
        ra, rb = a.enumerate(k), b.enumerate(k)
        assert ra[0] == rb[0] and (ra[1] == rb[1]).all() and (ra[2] == rb[2]).all()


def test_boost_order_is_not_insertion_order():
    keys = list(range(40))
    out = boost_order(keys)
    assert sorted(out) == keys and out != keys and out != keys[::-1]
