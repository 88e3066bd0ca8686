"""Hash-prefix sharded enumeration (csrc/shard.hip) on the single GPU of the test box:
several virtual ranks (one host thread each, local device-to-device transport) must reproduce the reference's
golden vectors bit for bit, and the RCCL transport is exercised with a one-rank communicator."""
import numpy as np
import pytest

from tests import vectors as V

pytestmark = pytest.mark.gpu

VECS = V.load_vectors()


def _bulges(v):
    return sum(o.get("bulges", 0) for o in v["outputs"])


def _max_k(v):
    return max(int(o["cmd"].split(":")[1]) for o in v["outputs"])


PICK = [v for v in VECS if (v["name"].startswith("hand/") or (v["name"].startswith("small/") and _bulges(v) < 60)) and _max_k(v) <= 32][:40]
GENOMES = [v for v in VECS if v["name"] in ("real/hpylori_k25", "real/hpylori_fine", "synth/strains4_100k", "synth/strains4_100k_fine")]


def _sharded(nranks):
    from sibelia_amd.dist import LocalShardedFinder
    return lambda seqs: LocalShardedFinder(seqs, [0] * nranks)


@pytest.mark.parametrize("v", PICK, ids=[v["name"] for v in PICK])
def test_sharded_matches_reference_small(v):
    V.replay(v, _sharded(3))


@pytest.mark.parametrize("nranks", [2, 5])
@pytest.mark.parametrize("v", GENOMES, ids=[v["name"] for v in GENOMES])
def test_sharded_matches_reference_genomes(v, nranks):
    V.replay(v, _sharded(nranks))


def test_sharded_equals_single_gpu_and_reports_exchange():
    from sibelia_amd import BlockFinder, workloads as W
    seqs = W.gen_strains(L0=200_000, n=4, seed=3, inv_min=5000, inv_max=20000)
    one = BlockFinder(seqs, device=0)
    many = _sharded(4)(seqs)
    a, b = one.enumerate(25), many.enumerate(25)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(one.list_edges(25), many.list_edges(25))
    assert one.simplify_stage(25, 150, 4) == many.simplify_stage(25, 150, 4)
    (sa, pa), (sb, pb) = one.state(), many.state()
    assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    st = many.stats()
    assert all(s["exchange_bytes"] > 0 for s in st)
    # every distinct k-mer of a slice leaves as ONE 16-B record: the all-to-all volume is bounded by 16 B per base position
    N = W.strand_kmers(seqs, 25)
    assert sum(s["exchange_bytes"] for s in st) < 16 * (N // 2) + 40 * st[0]["instances"] + (1 << 20)


def test_long_k_is_replicated_not_sharded():
    from sibelia_amd import BlockFinder, workloads as W
    seqs = W.gen_strains(L0=30_000, n=3, seed=4, inv_min=500, inv_max=2000)
    one, many = BlockFinder(seqs, device=0), _sharded(2)(seqs)
    a, b = one.enumerate(64), many.enumerate(64)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_rccl_transport_single_rank():
    # the RCCL code path (grouped ncclSend/ncclRecv to self + ncclAllGather) on the one GPU available here
    from sibelia_amd import BlockFinder, workloads as W
    from sibelia_amd.api import comm_unique_id
    seqs = W.gen_strains(L0=100_000, n=4, seed=8, inv_min=2000, inv_max=8000)
    one, rc = BlockFinder(seqs, device=0), BlockFinder(seqs, device=0)
    rc.attach_rccl(0, 1, comm_unique_id())
    a, b = one.enumerate(25), rc.enumerate(25)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert one.simplify_stage(25, 150, 4) == rc.simplify_stage(25, 150, 4)
    assert one.state()[0] == rc.state()[0]
    rc.detach()
    c = rc.enumerate(25)
    d = one.enumerate(25)
    assert c[0] == d[0] and np.array_equal(c[1], d[1])
