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


class HostFedInference:
    """Forward over images that live in HOST memory, with the upload of micro-batch i + 1 under the compute of
    micro-batch i: two pinned staging buffers and two device buffers, a copy stream beside the compute stream, events
    between them.  The caller sees ``run(images) -> outputs`` with ``images`` a CPU tensor (N, C, H, W), fp32 or bf16;
    ``compute_dtype`` (default bf16) is what the replica computes in, and an fp32 upload is cast on the device.
    Pinned, contiguous ``images`` are uploaded straight from where they are (leave them alone until the returned tensor
    has been synchronised with); pageable ones pass through the staging
    buffers (a host memcpy per micro-batch, which then sets the pace: ~150 MB per 256 fp32 images of 224 x 224).

        feeder = HostFedInference(model, device, micro_batch=256)
        logits = feeder.run(images_on_host)                # (N, classes) on the device, image order

    Bit-identical to ``model(images.to(device).to(compute_dtype))`` taken micro-batch by micro-batch: the kernels and
    their inputs are the same, only the copies overlap.  One instance serves one caller at a time.
    """

    def __init__(self, model, device, micro_batch: int = 256, compute_dtype=None):
        import torch
        self.model, self.device, self.micro_batch = model, torch.device(device), int(micro_batch)
        if self.micro_batch <= 0:
            raise ValueError("micro_batch must be positive")
        self.compute_dtype = torch.bfloat16 if compute_dtype is None else compute_dtype
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned = [None, None]
        self._dev = [None, None]
        self._ready = [torch.cuda.Event(), torch.cuda.Event()]     # upload into device buffer s finished
        self._free = [torch.cuda.Event(), torch.cuda.Event()]      # compute has consumed device buffer s
        self._staged = [torch.cuda.Event(), torch.cuda.Event()]    # pinned buffer s may be overwritten by the host
        self._used = [False, False]                                # slot s has events recorded (kept across run() calls:
                                                                   # the previous call's last uploads / forwards may still be in flight)

    def _buffers(self, like, rows: int, stage: bool) -> None:
        import torch
        shape = (rows,) + tuple(like.shape[1:])
        for s in range(2):
            if self._dev[s] is None or self._dev[s].shape != shape or self._dev[s].dtype != like.dtype:
                if self._used[s]:                  # the old buffers may still be read: let their last users finish first
                    self._staged[s].synchronize()
                    self._free[s].synchronize()
                    self._used[s] = False
                self._dev[s] = torch.empty(shape, dtype=like.dtype, device=self.device)
                self._pinned[s] = None
            if stage and self._pinned[s] is None:
                self._pinned[s] = torch.empty(shape, dtype=like.dtype, pin_memory=True)

    def run(self, images):
        import torch
        if images.is_cuda:
            raise ValueError("HostFedInference takes host tensors; call the model directly on device tensors")
        n = images.shape[0]
        if n == 0:
            return None
        mb = min(self.micro_batch, n)
        pinned_src = images.is_pinned() and images.is_contiguous()   # upload straight from the caller's pinned memory
        self._buffers(images, mb, stage=not pinned_src)
        compute = torch.cuda.current_stream(self.device)
        outs, used = [], self._used
        with torch.no_grad():
            for i, lo in enumerate(range(0, n, mb)):
                s, rows = i & 1, min(mb, n - lo)
                if pinned_src:
                    src = images[lo:lo + rows]
                else:
                    if used[s]:
                        self._staged[s].synchronize()          # the previous upload out of this pinned buffer is done
                    self._pinned[s][:rows].copy_(images[lo:lo + rows])     # pageable -> pinned (a host memcpy)
                    src = self._pinned[s][:rows]
                with torch.cuda.stream(self._copy_stream):
                    if used[s]:
                        self._copy_stream.wait_event(self._free[s])        # compute is done with device buffer s
                    self._dev[s][:rows].copy_(src, non_blocking=True)
                    self._staged[s].record(self._copy_stream)
                    self._ready[s].record(self._copy_stream)
                compute.wait_event(self._ready[s])
                x = self._dev[s][:rows]
                if x.dtype != self.compute_dtype:
                    x = _cast(x, self.compute_dtype)
                outs.append(self.model(x))
                self._free[s].record(compute)
                used[s] = True
        return outs[0] if len(outs) == 1 else torch.cat(outs)


def _cast(x, dtype):
    import torch
    from . import _ops
    if dtype == torch.bfloat16:
        return _ops.to_bf16(x)
    return x.to(dtype)
