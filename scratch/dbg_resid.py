import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops
torch.manual_seed(0)
M, N, K = 8192, 2304, 768
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = torch.randn(N, K, device="cuda") * K ** -0.5
for name, gam, bias, res in (("g1c0r0", 1.0, 0.0, 0.0), ("g1c0r1", 1.0, 0.0, 1.0), ("g2c1r0", 2.0, 1.0, 0.0)):
    pk = ops.pack_linear(W, torch.full((N,), bias, device="cuda"), out_scale=torch.full((N,), gam, device="cuda"))
    x = torch.full((M, N), res, device="cuda").to(torch.bfloat16)
    if res: x = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    ref = x.float() + gam * (A.float() @ W.to(torch.bfloat16).float().t() + bias)
    ops.gemm(A, pk, x, ops.EPI_RESIDUAL, M, resid=x)
    err = (x.float() - ref).abs()
    bad = err > 0.05
    print(name, "max err", err.max().item(), "bad frac", bad.float().mean().item())
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("  bad rows (first 20):", rows[:20].tolist(), "count", len(rows))
        print("  bad cols (first 40):", cols[:40].tolist(), "count", len(cols))
        r0 = rows[0].item()
        print("  row", r0, "bad cols:", bad[r0].nonzero().flatten()[:40].tolist())
        c0 = bad[r0].nonzero().flatten()[0].item()
        print("  got", x[r0, c0:c0+8].float().tolist(), "ref", ref[r0, c0:c0+8].tolist())
