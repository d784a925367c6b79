"""HIP-graph replay of a forward.

A forward of this package is a fixed sequence of kernel launches for a fixed input shape (the reference's
modules have no data-dependent control flow, vitm/models/models.py:208-211,352-361,408-416), so it can be
captured once into a HIP graph and replayed with one host call.  That matters where the launches themselves are
the cost: the small configurations (CIFAR ViT-tiny: ~90 launches of a few microseconds each per forward) and
host-bound multi-GPU driving.  The kernels are the same ones, on the same buffers: a replay is bit-identical to
the eager call.
"""
from __future__ import annotations

import torch


class GraphedForward:
    """``g = GraphedForward(module, example_input)`` then ``y = g(x)`` for any ``x`` of that shape and dtype.

    The returned tensor is the graph's static output buffer: it is overwritten by the next call (clone it to
    keep it).  Weights are read from the module's packed copies as they were at capture time; re-capture after
    changing them.
    """

    def __init__(self, module: torch.nn.Module, example: torch.Tensor, warmup: int = 2):
        if not example.is_cuda:
            raise RuntimeError("GraphedForward needs a CUDA example input")
        self.module = module
        self.static_in = example.detach().clone()
        with torch.no_grad():
            side = torch.cuda.Stream(device=example.device)
            side.wait_stream(torch.cuda.current_stream(example.device))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):   # packs the weights, opts the kernels into their LDS sizes
                    module(self.static_in)
            torch.cuda.current_stream(example.device).wait_stream(side)
            torch.cuda.synchronize(example.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = module(self.static_in)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise RuntimeError(f"captured for {tuple(self.static_in.shape)} {self.static_in.dtype}, "
                               f"got {tuple(x.shape)} {x.dtype}")
        if x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out


class GraphedTrainStep:
    """``Trainer.step`` replayed from a HIP graph: ``g = GraphedTrainStep(trainer)`` then ``loss = g.step(images, labels)``.

    A training step of this package is a fixed sequence of ~900 kernel launches for a fixed batch shape; at the
    reference's own batch sizes (32 .. 64 CIFAR images, vitm/config/config.yaml) the launches, not the kernels, are the
    step time.  The first ``warmup`` calls run ``trainer.step`` eagerly (real steps: they also opt the kernels into their
    LDS sizes); the next call captures the kernels of a step and from then on every call is: write the seven AdamW scalars
    for the current learning rate and step count to the device, copy the batch into the static input buffers, one graph
    launch, ``scheduler.step()``.  The kernels and their order are those of the eager step.  Needs the flat AdamW of this
    package, a single process (no gradient all-reduce) and batches of one shape; the returned loss is a copy.
    """

    def __init__(self, trainer, warmup: int = 2):
        if trainer.params is None or trainer.reducer.active:
            raise RuntimeError("GraphedTrainStep needs vitm_amd.training.AdamW in flat mode and a single process")
        self.trainer, self.warmup, self.calls = trainer, max(1, int(warmup)), 0
        self.graph = None

    def _capture(self, images, labels):
        tr = self.trainer
        dev = images.device
        self.x, self.y = images.detach().clone(), labels.detach().clone()
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.norm = tr.device_step(self.x, self.y, self.hyper)

    def step(self, images: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        tr = self.trainer
        self.calls += 1
        if self.calls <= self.warmup:
            return tr.step(images, labels)
        if self.graph is None:
            self._capture(images, labels)
        if images.shape != self.x.shape or images.dtype != self.x.dtype or labels.shape != self.y.shape:
            raise RuntimeError(f"captured for images {tuple(self.x.shape)} {self.x.dtype}, got {tuple(images.shape)} {images.dtype}")
        if not tr.optimizer._flat_ok():
            raise RuntimeError("the parameters or gradients left the flat buffers the graph was captured on")
        vals = tr.optimizer.next_hyper()
        # a pageable source: the copy is staged at once, so the next call may overwrite nothing that is still to be read
        self.hyper[:7].copy_(torch.tensor(vals, dtype=torch.float32))
        self.x.copy_(images, non_blocking=True)
        self.y.copy_(labels.to(self.y.dtype), non_blocking=True)
        self.graph.replay()
        tr.last_grad_norm = self.norm
        if tr.scheduler is not None:
            tr.scheduler.step()
        return self.loss.clone()

