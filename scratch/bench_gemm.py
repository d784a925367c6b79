"""Micro-benchmark of the GEMM shapes of ViT-B/16 (B=256) through the C ABI; prints TFLOP/s per shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")):
    sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops
torch.manual_seed(0)
DEV = "cuda"
B = int(os.environ.get("B", "256"))
M = B * 197
def run(name, N, K, epi, iters=30):
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    W = torch.randn(N, K, device=DEV) * K ** -0.5
    b = torch.randn(N, device=DEV) * 0.1
    if epi in (ops.EPI_LN, ops.EPI_LN_GELU):
        pk = ops.pack_linear(W, b, torch.ones(K, device=DEV), torch.zeros(K, device=DEV))
        stats = ops.ln_stats(A, M, K, 1e-5)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        if os.environ.get("FUSE", "0") == "1":
            f = lambda: ops.gemm_ln(A, pk, out, epi, M, K, 1e-5)
        else:
            f = lambda: ops.gemm(A, pk, out, epi, M, stats=stats)
    elif epi == ops.EPI_RESIDUAL:
        pk = ops.pack_linear(W, b, out_scale=torch.ones(N, device=DEV))
        x = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        f = lambda: ops.gemm(A, pk, x, epi, M, resid=x)
    else:
        pk = ops.pack_linear(W, b)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        f = lambda: ops.gemm(A, pk, out, epi, M)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:10s} M={M} N={N:5d} K={K:5d} epi={epi}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
ONLY = os.environ.get("ONLY")
_run = run
def run(name, *a, **k):
    if ONLY is None or name in ONLY.split(","): _run(name, *a, **k)
run("qkv", 2304, 768, ops.EPI_LN)
run("proj", 768, 768, ops.EPI_RESIDUAL)
run("fc1", 3072, 768, ops.EPI_LN_GELU)
run("fc2", 768, 3072, ops.EPI_RESIDUAL)
run("plain768", 768, 768, ops.EPI_BIAS)
run("plain3072", 3072, 768, ops.EPI_BIAS)
run("plainK3072", 768, 3072, ops.EPI_BIAS)
run("sq4096", 4096, 4096, ops.EPI_BIAS)
