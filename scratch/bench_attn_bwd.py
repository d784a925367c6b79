"""Time of the bf16 attention backward launch (ViT-B/16 shapes at batch 256), one line per library (VITM_GFX950_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops, _train_ops as T
torch.manual_seed(0)
B, N, H, hd = 256, 197, 12, 64
D = H * hd
qkv = (torch.randn(B * N, 3 * D, device="cuda")).to(torch.bfloat16)
o = torch.empty((B * N, D), dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * H * N, dtype=torch.float32, device="cuda")
T.attention_lse(qkv, o, lse, B, N, H, hd, hd ** -0.5)
do = (torch.randn(B * N, D, device="cuda")).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
f = lambda: T.attention_bwd(qkv, o, do, lse, dqkv, B, N, H, hd, hd ** -0.5)
for _ in range(5): f()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
print(os.path.basename(os.environ.get("VITM_GFX950_LIB", "default")), "attention backward: %.1f us" % best, flush=True)
