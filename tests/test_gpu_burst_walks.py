"""The walks that issue all their chunks at once (reservation: wave_core_claim_burst / wave_flank_order_burst, publish of a collapse:
wave_publish_collapse) must visit exactly what the step-wise walks visit -- same claims, same pushes, same stamps -- so a stage run with
them switched off (SBL_TEST_FLAGS=12) has to go through the same rounds and leave the same state.  The workload has indels and
inversions: walks that meet inserted elements or link breaks fall back to the step-wise form in the middle of the run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(seqs, k, D, it):
    from sibelia_amd import BlockFinder
    bf = BlockFinder(seqs, device=0)
    try:
        n = bf.simplify_stage(k, D, it)
        st = bf.stats()
        seq, pos = bf.state()
        return n, int(st["rounds"]), seq, pos
    finally:
        bf.close()


@pytest.mark.parametrize("k,D", [(25, 150), (15, 60), (31, 300)])
def test_burst_walks_change_nothing(monkeypatch, k, D):
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    seqs = W.gen_strains(L0=150_000, n=6, seed=21, inv_min=1000, inv_max=6000)
    a = _run(seqs, k, D, 4)
    monkeypatch.setenv("SBL_TEST_FLAGS", "12")
    b = _run(seqs, k, D, 4)
    monkeypatch.delenv("SBL_TEST_FLAGS")
    assert a[0] == b[0] and a[1] == b[1], "bulges / rounds differ between burst and step-wise walks"
    assert a[2] == b[2] and all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    orc = Oracle(seqs)
    assert orc.simplify_stage(k, D, 4) == a[0]
    so, po = orc.state()
    assert so == a[2] and all(np.array_equal(x, y) for x, y in zip(po, a[3]))
