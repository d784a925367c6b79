"""Micro-benchmark of the fused attention kernel through the C ABI."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops
torch.manual_seed(0)
def run(B, N, H, hd, iters=30):
    D = H * hd
    qkv = (torch.randn(B * N, 3 * D, device="cuda") * 1.5).to(torch.bfloat16)
    out = torch.empty((B * N, D), dtype=torch.bfloat16, device="cuda")
    f = lambda: ops.attention(qkv, out, B, N, H, hd, hd ** -0.5)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"B={B} N={N} H={H} hd={hd}: {us:7.1f} us  {4.0*B*H*N*N*hd/us/1e6:6.1f} TFLOP/s", flush=True)
run(256, 197, 12, 64)
run(64, 577, 12, 64)
run(256, 197, 6, 64)
run(256, 65, 3, 64)
