"""128x128-tile kernel against the 256x256 kernel on the ViT-S shapes with N = 384 (row_group = M forces the small kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops
torch.manual_seed(0)
def t(f, iters=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in [(50432, 384, 384), (50432, 384, 1536), (50432, 1152, 384), (50432, 1536, 384)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    g = torch.rand(N, device="cuda") + 0.5
    x = torch.randn(M + 2, N, device="cuda").to(torch.bfloat16)
    pk = ops.pack_linear(W, b, out_scale=g)
    big = t(lambda: ops.gemm(A, pk, x, ops.EPI_RESIDUAL, M, resid=x))
    small = t(lambda: ops.gemm(A, pk, x, ops.EPI_RESIDUAL, M, resid=x, row_group=M))
    fl = 2.0 * M * N * K
    print(f"M {M} N {N} K {K} residual epilogue: 256x256 {big:7.1f} us ({fl / big / 1e6:6.0f} TF)   128x128 {small:7.1f} us ({fl / small / 1e6:6.0f} TF)", flush=True)
