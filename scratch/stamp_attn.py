"""Per-phase timeline of the short-sequence attention kernel from a -DATTN_STAMP build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch, numpy as np
from vitm_amd import _ops as ops, _lib
lib = _lib.load()
B, N, H, hd = 256, 197, 12, 64
D = H * hd
qkv = (torch.randn(B * N, 3 * D, device="cuda") * 1.5).to(torch.bfloat16)
out = torch.empty((B * N, D), dtype=torch.bfloat16, device="cuda")
dbg = torch.zeros((256, 12, 8, 8), dtype=torch.int64, device="cuda")
for _ in range(5):
    _lib.check(lib.vitm_attention_lse(ops._p(qkv), 3 * D, ops._p(out), D, ops._p(dbg), B, N, H, hd, hd ** -0.5, ops._stream()), "attn")
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype("float64")[:, :, :7, :]     # [wg, item, wave, stamp]
names = ["issue dma/q/stores", "phase1 S", "phase2 softmax", "phase3 PV", "normalise+pack", "wait vmcnt", "barrier"]
for i, n in enumerate(names):
    seg = d[..., i + 1] - d[..., i]
    print(f"{n:22s} median {np.median(seg[:, 2:10]):8.0f}  p90 {np.percentile(seg[:, 2:10], 90):8.0f}  by wave {np.median(seg[:, 2:10], axis=(0, 1)).astype(int)}")
tot = d[:, 3:10, 0, 0] - d[:, 2:9, 0, 0]
print("cycles per pair (wave 0):", np.median(tot))
