"""One line per library (VITM_GFX950_LIB): best-of-3 time of the attention launch at the two benchmarked sequence lengths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops
torch.manual_seed(0)
def run(B, N, H, hd, iters=40):
    D = H * hd
    qkv = (torch.randn(B * N, 3 * D, device="cuda") * 1.5).to(torch.bfloat16)
    out = torch.empty((B * N, D), dtype=torch.bfloat16, device="cuda")
    f = lambda: ops.attention(qkv, out, B, N, H, hd, hd ** -0.5)
    for _ in range(10): f()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
only577 = len(sys.argv) > 1
print(os.path.basename(os.environ.get("VITM_GFX950_LIB", "default")), ("" if only577 else " N=197: %.1f us" % run(256, 197, 12, 64)) + "   N=577: %.1f us" % run(64, 577, 12, 64), flush=True)
