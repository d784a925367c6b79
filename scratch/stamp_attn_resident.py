"""Per-wave timeline of the resident-K/V attention kernel (577 tokens) from a -DATTN_STAMP build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch, numpy as np
from vitm_amd import _ops as ops, _lib
lib = _lib.load()
B, N, H, hd = 64, 577, 12, 64
D = H * hd
qkv = (torch.randn(B * N, 3 * D, device="cuda") * 1.5).to(torch.bfloat16)
out = torch.empty((B * N, D), dtype=torch.bfloat16, device="cuda")
dbg = torch.zeros((256, 4, 8, 3, 8), dtype=torch.int64, device="cuda")
for _ in range(5):
    _lib.check(lib.vitm_attention_lse(ops._p(qkv), 3 * D, ops._p(out), D, ops._p(dbg), B, N, H, hd, hd ** -0.5, ops._stream()), "attn")
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype("float64")     # [wg, unit iter, wave, block iter, slot]
u = d[:, 1]                                  # second unit of every workgroup
med = lambda x: np.median(x, axis=0).astype(int)
print("unit: load K/V (0->1)            by wave", med(u[:, :, 0, 1] - u[:, :, 0, 0]))
print("unit: barrier (1->2)             by wave", med(u[:, :, 0, 2] - u[:, :, 0, 1]))
for bi in range(3):
    print(f"block {bi}: strip 0 (3->4)        by wave", med(u[:, :, bi, 4] - u[:, :, bi, 3]))
    print(f"block {bi}: strip 1 (4->5)        by wave", med(u[:, :, bi, 5] - u[:, :, bi, 4]))
    print(f"block {bi}: strip 2, 5 tiles (5->6) by wave", med(u[:, :, bi, 6] - u[:, :, bi, 5]))
    print(f"block {bi}: normalise + store (6->7) by wave", med(u[:, :, bi, 7] - u[:, :, bi, 6]))
print("unit total (wave 0: start of unit 1 -> start of unit 2):", np.median(d[:, 2, 0, 0, 0] - d[:, 1, 0, 0, 0]))
print("wave end of last block relative to unit start, by wave:", med(np.max(u[:, :, :, 7], axis=2) - u[:, 0:1, 0, 0]))
