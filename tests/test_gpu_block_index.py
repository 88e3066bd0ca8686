"""The block index of the original slots (GraphView::bidx, round 5): the probe (k_probe_idx) and the reservation (reserve_idx_*) of an
ordered round read 32-byte records per 64-slot block instead of walking the windows element by element.  Two things must hold:
  * the index is KEPT UP TO DATE by the transactions (AddPoint / ErasePoint, link changes of a collapse, write stamps):
    SBL_CHECK_INDEX=1 compares it with a rebuild at the end of every stage and fails the stage on any difference;
  * reading it changes NOTHING: a stage with the index switched off (SBL_NO_BLOCK_INDEX=1: every window walked, as in round 4) goes
    through the same rounds and leaves the same state, which is the oracle's.
The workloads have indels and inversions (blocks that stop being pristine fall back to the walks in the middle of a run), many
chromosomes (separators inside windows), and a cascade of stages."""
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


def _same(a, b, what):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[1] == y[1], "bulges / rounds differ: " + what
        assert x[2] == y[2] and all(np.array_equal(p, q) for p, q in zip(x[3], y[3])), "state differs: " + what


@pytest.mark.parametrize("k,D", [(25, 150), (15, 60), (31, 300), (20, 500)])
def test_index_changes_nothing_and_stays_current(monkeypatch, k, D):
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    seqs = W.gen_strains(L0=150_000, n=6, seed=33, inv_min=1000, inv_max=6000)
    monkeypatch.setenv("SBL_CHECK_INDEX", "1")
    a = _run(seqs, [(k, D)])
    monkeypatch.setenv("SBL_NO_BLOCK_INDEX", "1")
    b = _run(seqs, [(k, D)])
    monkeypatch.delenv("SBL_NO_BLOCK_INDEX")
    _same(a, b, "indexed against walked")
    for sw in ("SBL_NO_IDX_PROBE", "SBL_NO_IDX_RESERVE"):                  # each reader alone
        monkeypatch.setenv(sw, "1")
        _same(a, _run(seqs, [(k, D)]), sw)
        monkeypatch.delenv(sw)
    orc = Oracle(seqs)
    assert orc.simplify_stage(k, D, 4) == a[0][0]
    so, po = orc.state()
    assert so == a[0][2] and all(np.array_equal(x, y) for x, y in zip(po, a[0][3]))


def test_index_over_a_cascade_with_many_chromosomes(monkeypatch):
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    base = W.gen_strains(L0=60_000, n=5, seed=7, inv_min=500, inv_max=3000)
    # cut every strain into pieces of a few kbp: separators inside most neighbourhoods, chromosomes shorter than a window among them
    rng = np.random.default_rng(5)
    seqs = []
    for s in base:
        p = 0
        while p < len(s):
            q = min(len(s), p + int(rng.integers(40, 6000)))
            seqs.append(s[p:q]); p = q
    stages = [(22, 100), (25, 150), (30, 150)]
    monkeypatch.setenv("SBL_CHECK_INDEX", "1")
    a = _run(seqs, stages)
    monkeypatch.setenv("SBL_NO_BLOCK_INDEX", "1")
    b = _run(seqs, stages)
    monkeypatch.delenv("SBL_NO_BLOCK_INDEX")
    _same(a, b, "indexed against walked, cascade")
    orc = Oracle(seqs)
    for (k, D), x in zip(stages, a):
        assert orc.simplify_stage(k, D, 4) == x[0]
        so, po = orc.state()
        assert so == x[2] and all(np.array_equal(p, q) for p, q in zip(po, x[3]))


def test_index_survives_checkpointed_replays(monkeypatch):
    """an attempt that is abandoned and rerun with iteration checkpoints (pool too small: grow + replay) rebuilds the index from the restored arrays"""
    from sibelia_amd import workloads as W
    seqs = W.gen_strains(L0=80_000, n=6, seed=12, inv_min=1000, inv_max=4000)
    a = _run(seqs, [(25, 150)])
    monkeypatch.setenv("SBL_CHECK_INDEX", "1")
    monkeypatch.setenv("SBL_TEST_ELEM_SLACK", "64")
    b = _run(seqs, [(25, 150)])
    assert a[0][0] == b[0][0] and a[0][2] == b[0][2] and all(np.array_equal(p, q) for p, q in zip(a[0][3], b[0][3]))
