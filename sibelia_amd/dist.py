"""Multi-GPU job layout (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

Bulge removal is globally ordered, so in this round the stage does not shard: every rank runs the whole
hot path on its own input ("replicas only", weak scaling, no data-path collective).  The only collectives
are the barrier around the timed region and two scalar all-reduces (max time, total units)."""
from __future__ import annotations

from typing import List, Tuple


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
