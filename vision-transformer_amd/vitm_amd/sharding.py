"""Batch sharding for multi-GPU inference: one process per GPU, a full weight replica each, contiguous image
chunks, no collective on the data path (images are independent through the whole forward)."""
from __future__ import annotations

from typing import List, Tuple


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of rank's contiguous chunk of n_items; the first n_items % world_size ranks get one extra."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} for world size {world_size}")
    if n_items < 0:
        raise ValueError("n_items must be >= 0")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_shard_bounds(n_items: int, world_size: int) -> List[Tuple[int, int]]:
    return [shard_bounds(n_items, world_size, r) for r in range(world_size)]


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """MAX-reduce a per-rank scalar (elapsed seconds) over the process group; identity without one."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
