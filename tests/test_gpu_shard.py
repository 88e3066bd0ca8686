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


LONGK = [v for v in VECS if v["name"] in ("real/hpylori_loose", "real/hpylori_fine", "synth/strains4_100k_fine")]
LONGK_SMALL = [v for v in VECS if v["name"].startswith("small/") and _max_k(v) > 32 and _bulges(v) < 400][:12]


@pytest.mark.parametrize("nranks", [2, 3, 5])
@pytest.mark.parametrize("v", LONGK, ids=[v["name"] for v in LONGK])
def test_sharded_long_k_matches_reference_genomes(v, nranks, monkeypatch):
    """k > 32 split over the attached GPUs: the -s loose / -s fine cascades of the reference (k = 100 .. 5000) through 2, 3 and 5
    virtual ranks, every output identical on every rank -- by the fingerprint table sharded by hash prefix (csrc/longk_fp.hip, the
    default since round 6) and, with three ranks, also by the sharded rank doubling (csrc/longk.hip: SBL_LONGK_DOUBLING=1, the fall-back)"""
    V.replay(v, _sharded(nranks))
    if nranks == 3:
        monkeypatch.setenv("SBL_LONGK_DOUBLING", "1")
        V.replay(v, _sharded(nranks))


@pytest.mark.parametrize("v", LONGK_SMALL, ids=[v["name"] for v in LONGK_SMALL])
def test_sharded_long_k_matches_reference_small(v):
    # tiny inputs: slices shorter than the halo, empty slices, records shorter than k
    V.replay(v, _sharded(3))
    V.replay(v, _sharded(5))


def test_long_k_is_sharded_not_replicated(monkeypatch):
    """With a communicator attached the k > 32 enumeration is ONE distributed job: every rank sorts about 1 / R of the suffixes and the
    ranks exchange 12 + 8 B per active suffix and round; SBL_LONGK_REPLICATED=1 brings back the replicated run (same result, no bytes)."""
    from sibelia_amd import BlockFinder, workloads as W
    seqs = W.gen_strains(L0=30_000, n=3, seed=4, inv_min=500, inv_max=2000)
    one, many = BlockFinder(seqs, device=0), _sharded(4)(seqs)
    monkeypatch.setenv("SBL_LONGK_DOUBLING", "1")      # (the exchange bound below is the sharded rank doubling's; the fingerprint table's sharding: tests/test_gpu_longk_fp.py)
    for k in (33, 64, 100, 500):
        a, b = one.enumerate(k), many.enumerate(k)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        st = many.stats()
        assert all(s["exchange_bytes"] > 0 for s in st), "the long-k enumeration did not exchange anything: replicated?"
        nsuf = 2 * (sum(len(s) for s in seqs) + len(seqs) + 1) - 1 + k
        rounds = max(1, int(np.log2(k)) - 3)
        assert sum(s["exchange_bytes"] for s in st) < (20 * rounds + 12 + 8) * nsuf + (1 << 16)      # 20 B per suffix and round at most
    monkeypatch.delenv("SBL_LONGK_DOUBLING")
    assert one.simplify_stage(100, 500, 4) == many.simplify_stage(100, 500, 4)
    (sa, pa), (sb, pb) = one.state(), many.state()
    assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    monkeypatch.setenv("SBL_LONGK_REPLICATED", "1")
    a, b = one.enumerate(64), many.enumerate(64)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert all(s["exchange_bytes"] == 0 for s in many.stats())


def test_sharded_config5_shape_matches_oracle():
    # config 5 shape (random DNA in 4 records, k = 5000: nearly every suffix is unique after h = 16) through 3 virtual ranks
    from sibelia_amd import workloads as W
    from oracle.oracle import Oracle
    seqs = W.longk_case(4_000_000, 4)
    many, orc = _sharded(3)(seqs), Oracle(seqs)
    a, b = many.enumerate(5000), orc.enumerate(5000)
    assert a[0] == b[0] and (a[1] == b[1]).all() and (a[2] == b[2]).all()
    assert many.simplify_stage(5000, 15000, 4) == orc.simplify_stage(5000, 15000, 4) == 1
    (sa, pa), (sb, pb) = many.state(), orc.state()
    assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))


def test_rccl_transport_single_rank_long_k():
    # the RCCL code path of the sharded rank doubling (grouped ncclSend / ncclRecv to self) on the one GPU available here
    from sibelia_amd import BlockFinder, workloads as W
    from sibelia_amd.api import comm_unique_id
    seqs = W.gen_strains(L0=60_000, n=4, seed=8, inv_min=2000, inv_max=8000)
    one, rc = BlockFinder(seqs, device=0), BlockFinder(seqs, device=0)
    rc.attach_rccl(0, 1, comm_unique_id())
    a, b = one.enumerate(100), rc.enumerate(100)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    rc.detach()


def test_read_only_simplification_phases_are_split_over_the_ranks(monkeypatch):
    """SURVEY.md 8e, row "Simplification": the commits of the ordered rounds stay replicated (bit-identical state everywhere), the
    read-only phases -- the AnyBulges snapshots of an iteration and the probe of every round -- are shared out: each rank takes the
    verdicts of its share of the ids / of the window and the verdict bytes are all-gathered (1 B per id per snapshot, 1 B per window
    entry per round).  Same result as one GPU, on every rank; SBL_REPLICATED_PHASES=1 brings back round 3's behaviour."""
    from sibelia_amd import BlockFinder, workloads as W
    seqs = W.gen_strains(L0=300_000, n=6, seed=5, inv_min=3000, inv_max=12000)
    one, many = BlockFinder(seqs, device=0), _sharded(4)(seqs)
    one.save_state()
    for bf in many.ranks:
        bf.save_state()
    want = one.simplify_stage(25, 150, 4)
    assert many.simplify_stage(25, 150, 4) == want
    (sa, pa), (sb, pb) = one.state(), many.state()
    assert sa == sb and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    st = many.stats()
    assert all(s["ro_ranks"] == 4 and s["verdict_bytes"] > 0 and s["replays"] == 0 for s in st)
    assert all(s["rounds"] == st[0]["rounds"] and s["transactions"] == st[0]["transactions"] for s in st)
    # 1 B per id and snapshot + 1 B per window entry and round, sent to 3 peers
    nid, rounds, iters = st[0]["bif_count"], st[0]["rounds"], st[0]["iterations"]
    assert sum(s["verdict_bytes"] for s in st) <= 3 * (iters * (nid + 4) + rounds * (4 * 14336 + 4 * 4 + 4))
    monkeypatch.setenv("SBL_REPLICATED_PHASES", "1")
    for bf in many.ranks:
        bf.restore_state()
    assert many.simplify_stage(25, 150, 4) == want
    assert many.state()[0] == sa
    assert all(s["ro_ranks"] == 1 and s["verdict_bytes"] == 0 for s in many.stats())


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


def _rccl_rank(rank, world, port, q):
    # one process per GPU: torch.distributed (backend "nccl" = RCCL) only carries the 128-byte communicator id,
    # the exchange itself is csrc/shard.hip's grouped ncclSend / ncclRecv + ncclAllGather
    import os
    import hashlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from sibelia_amd import BlockFinder, workloads as W, dist as D, formats as F
    seqs = W.gen_strains(L0=150_000, n=4, seed=12, inv_min=3000, inv_max=12000)
    bf = BlockFinder(seqs, device=rank)
    D.attach(bf, device=torch.device("cuda", rank))
    bc, pos, neg = bf.enumerate(25)
    cols = lambda a: np.stack([a["id"], a["chr"], a["pos"]], 1).astype("<u4") if len(a) else np.zeros((0, 3), "<u4")
    d_enum = hashlib.sha256(F.enum_bytes(bc, cols(pos), cols(neg))).hexdigest()
    bulges = bf.simplify_stage(25, 150, 4)
    s, p = bf.state()
    d_state = hashlib.sha256(F.state_bytes(bulges, s, p)).hexdigest()
    st = bf.stats()
    q.put((rank, d_enum, d_state, st["exchange_bytes"]))
    dist.barrier()
    bf.close()
    dist.destroy_process_group()


def _run_rccl(world):
    import hashlib
    import os
    import multiprocessing as mp          # (not torch.multiprocessing: this process must not load a second copy of librccl, see shard.hip)
    from sibelia_amd import BlockFinder, workloads as W, formats as F
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_rccl_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # the single-GPU result of the same input
    seqs = W.gen_strains(L0=150_000, n=4, seed=12, inv_min=3000, inv_max=12000)
    one = BlockFinder(seqs, device=0)
    bc, pos, neg = one.enumerate(25)
    cols = lambda a: np.stack([a["id"], a["chr"], a["pos"]], 1).astype("<u4") if len(a) else np.zeros((0, 3), "<u4")
    d_enum = hashlib.sha256(F.enum_bytes(bc, cols(pos), cols(neg))).hexdigest()
    bulges = one.simplify_stage(25, 150, 4)
    s, p = one.state()
    d_state = hashlib.sha256(F.state_bytes(bulges, s, p)).hexdigest()
    for rank, e, t, xb in res:
        assert e == d_enum and t == d_state, "rank %d disagrees with the single-GPU result" % rank
        assert world == 1 or xb > 0
    return res


def test_rccl_launch_harness_one_process():
    # the same one-process-per-GPU harness as the multi-rank test below, with world size 1 (always runnable)
    _run_rccl(1)


def test_rccl_all_to_all_across_all_gpus_of_the_node():
    # the real thing: one rank per visible GPU, k-mer records exchanged over xGMI; skipped on a single-GPU box
    import subprocess, sys
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    n = min(int(out.stdout.strip() or 0), 8)
    if n < 2:
        pytest.skip("needs at least two GPUs (this box has %d)" % n)
    _run_rccl(n)
