"""Calibration only: hipBLASLt (torch.matmul) on the ViT-B GEMM shapes, to know what the library reaches."""
import torch, time
M = 50432
for (N, K) in [(2304, 768), (3072, 768), (768, 768), (768, 3072), (4096, 4096)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.matmul(a, w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"N={N} K={K}: {us:.1f} us  {2*M*N*K/us/1e6:.0f} TF", flush=True)
