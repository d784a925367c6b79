"""Micro-benchmark of the weight-gradient GEMM (vitm_gemm_tn) on the ViT-B/16 shapes; prints TFLOP/s per shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")):
    sys.path.insert(0, p)
import torch
from vitm_amd import _train_ops as T
DEV = "cuda"
B = int(os.environ.get("B", "64"))
M = B * 197
ITERS = int(os.environ.get("ITERS", "20"))
for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    a = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    b = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    out = torch.empty((N, K), device=DEV)
    for _ in range(3): T.gemm_tn(a, b, M, N, K, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): T.gemm_tn(a, b, M, N, K, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / ITERS
    print(f"{name:5s} M={M} N={N:5d} K={K:5d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
