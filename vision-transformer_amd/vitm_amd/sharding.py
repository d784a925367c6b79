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


class ShardedInference:
    """Batch-sharded inference over the ranks of a process group: one replica per rank (GPU), each rank runs its
    contiguous chunk of a global batch, nothing is exchanged on the data path.

        runner = ShardedInference(model, dist)            # every rank, same model weights
        local, (start, stop) = runner(global_batch)       # this rank's outputs for images [start, stop)
        full = runner.gather(local)                       # optional: every rank's outputs on rank 0 (None elsewhere)

    ``global_batch`` is the same tensor on every rank (host or device memory; only this rank's rows are moved to
    the replica's device) or a callable ``(start, stop) -> tensor`` that produces just the rank's rows, so a loader
    never has to materialise the other ranks' images.  ``micro_batch`` bounds the rows per forward call.
    """

    def __init__(self, model, dist=None, device=None, micro_batch: int = 0):
        self.model, self.dist, self.device, self.micro_batch = model, dist, device, int(micro_batch)
        active = dist is not None and dist.is_initialized()
        self.world = dist.get_world_size() if active else 1
        self.rank = dist.get_rank() if active else 0

    def bounds(self, n_items: int) -> Tuple[int, int]:
        return shard_bounds(n_items, self.world, self.rank)

    def __call__(self, global_batch, n_items: int = None):
        import torch
        if callable(global_batch):
            if n_items is None:
                raise ValueError("n_items is required when the batch is produced by a callable")
            s, e = self.bounds(n_items)
            mine = global_batch(s, e)
        else:
            s, e = self.bounds(global_batch.shape[0])
            mine = global_batch[s:e]
        if self.device is not None:
            mine = mine.to(self.device, non_blocking=True)
        step = self.micro_batch if self.micro_batch > 0 else max(1, e - s)
        outs = []
        with torch.no_grad():
            for i in range(0, e - s, step):
                outs.append(self.model(mine[i:i + step]))
        if not outs:   # more ranks than images: this rank has nothing to do
            return None, (s, e)
        return (outs[0] if len(outs) == 1 else torch.cat(outs)), (s, e)

    def gather(self, local, dst: int = 0):
        """Outputs of all ranks, in image order, on rank ``dst`` (None on the others).  A convenience for callers
        that want one tensor; it is outside the data path and uses an object gather (host memory)."""
        import torch
        if self.world == 1:
            return local
        parts = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(None if local is None else local.detach().cpu(), parts, dst=dst)
        if self.rank != dst:
            return None
        return torch.cat([p for p in parts if p is not None])
