"""Multi-GPU job layout (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

Two ways to use N GPUs:
  * replicas (bench.py default for N > 1): bulge removal is globally ordered and is >90 % of a stage, so
    throughput scales by giving every GPU its own input; the only collectives are the barrier around the
    timed region and two scalar all-reduces (max time, total units).
  * sharded enumeration (csrc/shard.hip): the k-mer table of ONE job is sharded by hash prefix
    over the GPUs (all-to-all of 16-B k-mer records over RCCL/xGMI, all-gather of bifurcation codes and
    marks); simplification then runs replicated and bit-identical on every GPU.  `attach` wires a
    BlockFinder to the torch.distributed world; `LocalShardedFinder` drives several GPUs (or several
    virtual ranks on one GPU, as the tests do) from one process, one host thread per rank."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def rank_workload(rank: int, strains: int, L0: int) -> dict:
    """Arguments of workloads.gen_strains for this rank: same generator, rank-specific seed."""
    return {"L0": L0, "n": strains, "seed": 1 + rank}


def aggregate(dt: float, units: float, device=None) -> Tuple[float, float]:
    """(max over ranks of dt, sum over ranks of units); identity when not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dt, units
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def share_unique_id(make_id, device=None) -> bytes:
    """Rank 0 creates the 128-byte communicator id, everyone receives it (torch.distributed broadcast)."""
    import torch
    import torch.distributed as dist
    from .api import COMM_ID_BYTES
    if dist.get_rank() == 0:
        t = torch.frombuffer(bytearray(make_id()), dtype=torch.uint8).clone()
    else:
        t = torch.zeros(COMM_ID_BYTES, dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=0)
    return bytes(t.cpu().numpy().tobytes())


def attach(bf, device=None) -> None:
    """Attach an RCCL communicator spanning the torch.distributed world to this rank's BlockFinder."""
    import torch.distributed as dist
    from .api import comm_unique_id
    uid = share_unique_id(comm_unique_id, device)
    bf.attach_rccl(dist.get_rank(), dist.get_world_size(), uid)


class LocalShardedFinder:
    """One job on `len(devices)` ranks of ONE process (a host thread per rank, device-to-device exchange).

    Every rank holds the whole state; the enumeration is sharded, everything else replicated.  Results of all
    ranks must be identical -- that is asserted on every call, rank 0's result is returned."""

    def __init__(self, seqs: Sequence[bytes], devices: Sequence[int]):
        from .api import BlockFinder, LocalGroup
        self.group = LocalGroup(len(devices))
        self.ranks: List = [BlockFinder(seqs, device=d) for d in devices]
        for r, bf in enumerate(self.ranks):
            bf.attach_local(self.group, r)

    def _all(self, name, *a):
        return self.group.run([lambda bf=bf: getattr(bf, name)(*a) for bf in self.ranks])

    def enumerate(self, k):
        import numpy as np
        res = self._all("enumerate", k)
        for x in res[1:]:
            assert x[0] == res[0][0] and np.array_equal(x[1], res[0][1]) and np.array_equal(x[2], res[0][2]), "ranks disagree"
        return res[0]

    def simplify_stage(self, k, min_branch, max_iter):
        res = self._all("simplify_stage", k, min_branch, max_iter)
        assert all(x == res[0] for x in res), "ranks disagree"
        return res[0]

    PerformGraphSimplifications = simplify_stage

    def state(self):
        import numpy as np
        res = [bf.state() for bf in self.ranks]
        for s, p in res[1:]:
            assert s == res[0][0] and all(np.array_equal(x, y) for x, y in zip(p, res[0][1])), "ranks disagree"
        return res[0]

    def list_edges(self, k):
        import numpy as np
        res = self._all("list_edges", k)
        assert all(np.array_equal(x, res[0]) for x in res), "ranks disagree"
        return res[0]

    def generate_blocks(self, k, trim_k, min_size, shared_only=False):
        import numpy as np
        res = self._all("generate_blocks", k, trim_k, min_size, shared_only)
        assert all(np.array_equal(x, res[0]) for x in res), "ranks disagree"
        return res[0]

    def postprocess(self, names=None, glue=True):
        return self.ranks[0].postprocess(names, glue)

    def serialize_graph(self, k):
        return self.ranks[0].serialize_graph(k)

    def kmer_hashes(self, k):
        return self.ranks[0].kmer_hashes(k)

    def stats(self):
        return [bf.stats() for bf in self.ranks]

    def close(self):
        for bf in self.ranks:
            bf.close()
