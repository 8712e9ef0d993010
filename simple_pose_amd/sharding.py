"""Batch sharding across ranks (one process per GPU) and whole-job throughput aggregation.

The forward + decode path shards by image with no data-path collective; the only traffic is the
barrier and two scalar reductions of the measurement itself.  `rank_indices` is the reference's sampler rule
(`processors/ddp_pose_resnet_solver.py:42-48`: DistributedSampler without shuffle/set_epoch -> sample i goes to rank
i mod W, padded by wrap-around, then BatchSampler(drop_last=True)).
"""
from __future__ import annotations

from typing import List, Optional


def rank_indices(n_samples: int, rank: int, world: int, batch_size: Optional[int] = None) -> List[int]:
    """Indices rank `rank` of `world` processes, DistributedSampler(shuffle=False) semantics (pad by wrapping so every
    rank gets ceil(n/world) samples), then whole batches only when `batch_size` is given (drop_last=True)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    per_rank = -(-n_samples // world)
    padded = list(range(n_samples))
    padded += padded[: per_rank * world - n_samples]
    mine = padded[rank::world]
    if batch_size:
        mine = mine[: (len(mine) // batch_size) * batch_size]
    return mine


def aggregate_throughput(units_local: float, elapsed_local: float, device=None):
    """(total units over all ranks, MAX elapsed over ranks, units/s).  Works without an initialised process group."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return units_local, elapsed_local, units_local / elapsed_local
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    u = torch.tensor([units_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item()), float(u.item() / t.item())
