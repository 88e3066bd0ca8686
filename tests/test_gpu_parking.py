"""Parked transactions (GraphView::park_of, round 5): a transaction that has made SBL_PARK collapses in one launch of k_commit and has
decided another one parks -- its LDS state goes to the end of its arena slice, the id stays pending -- and k_resume takes it up in a
later round.  Parking must change NOTHING but the number of rounds: every setting (0 = off, 1 = park after every collapse, 2 = the
default, 3) leaves the state the oracle leaves.  The workloads have multi-collapse transactions (several strains with their own SNPs at
the same place), indels (non-pristine blocks, deferred Cleanup visible to the probes of neighbouring ids), many chromosomes, wide
windows (the walking probe for every entry: the arena slice of a parked transaction must survive it), replays with checkpoints, and
the many-instances regime (mark lists and AnyBulges tables in the arena)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(seqs, stages):
    from sibelia_amd import BlockFinder
    bf = BlockFinder(seqs, device=0)
    try:
        out = []
        for k, D in stages:
            n = bf.simplify_stage(k, D, 4)
            st = bf.stats()
            seq, pos = bf.state()
            out.append((n, int(st["rounds"]), seq, pos))
        return out
    finally:
        bf.close()


def _same_state(a, b, what):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x[0] == y[0], "bulges differ: " + what
        assert x[2] == y[2] and all(np.array_equal(p, q) for p, q in zip(x[3], y[3])), "state differs: " + what


def _oracle(seqs, stages):
    from oracle.oracle import Oracle
    orc = Oracle(seqs)
    out = []
    for k, D in stages:
        n = orc.simplify_stage(k, D, 4)
        so, po = orc.state()
        out.append((n, 0, so, po))
    return out


@pytest.mark.parametrize("k,D", [(25, 150), (15, 60), (31, 300), (20, 500)])
def test_parking_changes_nothing(monkeypatch, k, D):
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=120_000, n=8, seed=41, snp=0.03, inv_min=1000, inv_max=6000)
    ref = _oracle(seqs, [(k, D)])
    rounds = {}
    for cap in ("0", "1", "2", "3"):
        monkeypatch.setenv("SBL_PARK", cap)
        monkeypatch.setenv("SBL_CHECK_INDEX", "1")
        a = _run(seqs, [(k, D)])
        _same_state(a, ref, "SBL_PARK=" + cap)
        rounds[cap] = a[0][1]
    assert rounds["1"] > rounds["0"], "SBL_PARK=1 parked nothing (%s): the test does not exercise what it is for" % rounds


def test_parking_over_a_cascade_with_many_chromosomes(monkeypatch):
    from sibelia_amd import workloads as W
    base = W.gen_strains(L0=60_000, n=6, seed=9, snp=0.03, inv_min=500, inv_max=3000)
    rng = np.random.default_rng(6)
    seqs = []
    for s in base:
        p = 0
        while p < len(s):
            q = min(len(s), p + int(rng.integers(40, 6000)))
            seqs.append(s[p:q]); p = q
    stages = [(22, 100), (25, 150), (30, 150)]
    ref = _oracle(seqs, stages)
    for cap in ("1", "2"):
        monkeypatch.setenv("SBL_PARK", cap)
        _same_state(_run(seqs, stages), ref, "cascade, SBL_PARK=" + cap)


def test_parking_with_walking_probes_and_reservations(monkeypatch):
    """block index off: every probe walks in the arena slice of its window position -- the spare slice where a parked transaction lives"""
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=100_000, n=6, seed=13, snp=0.03, inv_min=1000, inv_max=4000)
    ref = _oracle(seqs, [(25, 150)])
    monkeypatch.setenv("SBL_NO_BLOCK_INDEX", "1")
    for cap in ("1", "2"):
        monkeypatch.setenv("SBL_PARK", cap)
        _same_state(_run(seqs, [(25, 150)]), ref, "walking probes, SBL_PARK=" + cap)


def test_parking_survives_checkpointed_replays(monkeypatch):
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=80_000, n=6, seed=12, snp=0.03, inv_min=1000, inv_max=4000)
    ref = _oracle(seqs, [(25, 150)])
    monkeypatch.setenv("SBL_PARK", "1")
    monkeypatch.setenv("SBL_TEST_ELEM_SLACK", "64")
    _same_state(_run(seqs, [(25, 150)]), ref, "grow + replay, SBL_PARK=1")


def test_parking_with_dozens_of_instances(monkeypatch):
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=8_000, n=48, seed=5, inv_min=100, inv_max=400)
    ref = _oracle(seqs, [(25, 150)])
    for cap in ("1", "2"):
        monkeypatch.setenv("SBL_PARK", cap)
        _same_state(_run(seqs, [(25, 150)]), ref, "48 strains, SBL_PARK=" + cap)
