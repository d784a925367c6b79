"""Whole batch on one stream against the same batch as two halves in flight on two streams (the tail of one stream's
persistent GEMM could be filled by the other stream's kernels).  python scratch/bench_two_streams.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import weights as W  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
cfg = W.CONFIGS["vit_b16"]
model, _ = bench.build_model(cfg, dev)
x = (torch.rand((B, 3, 224, 224)) * 2 - 1).to(dev).to(torch.bfloat16)
halves = [x[: B // 2].contiguous(), x[B // 2:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def whole():
    return model(x)


def split():
    cur = torch.cuda.current_stream()
    outs = []
    for s, h in zip(streams, halves):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(model(h))
    for s in streams:
        cur.wait_stream(s)
    return outs


def rate(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)


with torch.no_grad():
    for _ in range(150):
        whole()
    for rep in range(3):
        print(f"whole batch, one stream   {rate(whole, 40):9.0f} img/s", flush=True)
        print(f"two halves, two streams   {rate(split, 40):9.0f} img/s", flush=True)
    a = whole().float()
    b = torch.cat([o.float() for o in split()])
    torch.cuda.synchronize()
    print("max |whole - split| =", (a - b).abs().max().item())
