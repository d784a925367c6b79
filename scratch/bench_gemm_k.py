"""Time of the big-tile GEMM against K at fixed M, N: the intercept is the per-tile prologue + epilogue cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")):
    sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops
DEV = "cuda"
M = 256 * 197
N = int(os.environ.get("N", "3072"))
for epi_name, epi in (("bias", ops.EPI_BIAS), ("bias_gelu", ops.EPI_BIAS_GELU)):
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        W = torch.randn(N, K, device=DEV) * K ** -0.5
        pk = ops.pack_linear(W, torch.randn(N, device=DEV) * 0.1)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        f = lambda: ops.gemm(A, pk, out, epi, M)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{epi_name:9s} N={N} K={K:5d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
