"""N > 1 job layout on CPU: world_size 2 over gloo (the GPU path uses the same code over RCCL)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sibelia_amd import dist as D, workloads as W
    kw = D.rank_workload(rank, strains=2, L0=3000)
    seqs = W.gen_strains(inv_min=100, inv_max=400, **kw)
    units = float(W.strand_kmers(seqs, 25))
    dist.barrier()
    dt, total = D.aggregate(0.5 + rank, units)
    # the communicator id of the sharded enumeration travels from rank 0 to everyone (RCCL's id is opaque bytes)
    uid = D.share_unique_id(lambda: bytes(range(128)))
    assert uid == bytes(range(128))
    q.put((rank, kw["seed"], W.input_digest(seqs), units, dt, total))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_layout_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, d0, u0, t0, tot0), (r1, s1, d1, u1, t1, tot1) = res
    assert (s0, s1) == (1, 2) and d0 != d1                 # every rank owns a different strain set
    assert t0 == t1 == 1.5                                  # max over ranks
    assert tot0 == tot1 == u0 + u1                          # whole-job units


def test_aggregate_is_identity_without_process_group():
    from sibelia_amd import dist as D
    assert D.aggregate(2.0, 7.0) == (2.0, 7.0)
