"""Does the row stride of q|k|v matter to the 197-token attention kernel?  Same pairs, same bytes, two layouts:
(a) 256 images x 12 heads in [M, 2304] rows (an operand row of a pair every 4608 bytes), (b) 3072 one-head images in
[M, 192] rows (every 384 bytes: a pair's operands almost contiguous).  A cold-cache variant rotates over 6 buffers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops
torch.manual_seed(0)
def run(B, N, H, hd, nbuf, iters=48):
    D = H * hd
    qs = [(torch.randn(B * N, 3 * D, device="cuda") * 1.5).to(torch.bfloat16) for _ in range(nbuf)]
    outs = [torch.empty((B * N, D), dtype=torch.bfloat16, device="cuda") for _ in range(nbuf)]
    def f(i): ops.attention(qs[i % nbuf], outs[i % nbuf], B, N, H, hd, hd ** -0.5)
    for i in range(12): f(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): f(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
for nbuf in (1, 6):
    print(f"buffers {nbuf}: [M, 2304] rows (256 x 12 heads): {run(256, 197, 12, 64, nbuf):.1f} us   [M, 192] rows (3072 x 1 head): {run(3072, 197, 1, 64, nbuf):.1f} us", flush=True)
