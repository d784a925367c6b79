"""Per-phase timeline of the chunked-key attention kernel (577 tokens) from a -DATTN_STAMP build."""
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
dbg = torch.zeros((256, 16, 8, 8), dtype=torch.int64, device="cuda")
for _ in range(5):
    _lib.check(lib.vitm_attention_lse(ops._p(qkv), 3 * D, ops._p(out), D, ops._p(dbg), B, N, H, hd, hd ** -0.5, ops._stream()), "attn")
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype("float64")[:, :, :7, :]     # [wg, step, wave, stamp]
names = ["S phase", "(flush) + max", "exp + PV (+rescale)", "barrier wait"]
for c in range(3):
    print(f"chunk {c} of an item (steps {c + 3}, {c + 6}, {c + 9}):")
    sel = d[:, [c + 3, c + 6, c + 9]]
    for i, n in enumerate(names):
        seg = sel[..., i + 1] - sel[..., i]
        print(f"   {n:22s} median {np.median(seg):8.0f}  by wave {np.median(seg, axis=(0, 1)).astype(int)}")
tot = d[:, 4:13, 0, 0] - d[:, 3:12, 0, 0]
print("cycles per step (wave 0), by step:", np.median(tot, axis=0).astype(int))
print("steps 3..12 total:", np.median(d[:, 12, 0, 0] - d[:, 3, 0, 0]))

# item boundary: barrier exit of an item's last chunk (stamp 4 of step 3c+2) -> Q fragments in registers (5, next step)
# -> first chunk starts (0); and inside the first chunk: S phase end (1) -> previous output flushed (6)
a = d[:, [5, 8]][..., 4]
b = d[:, [6, 9]]
for n, x, y in (("barrier exit -> Q fragments in registers", a, b[..., 5]), ("-> chunk start", b[..., 5], b[..., 0]),
                ("chunk 0: S phase end -> output flushed", b[..., 1], b[..., 6])):
    seg = y - x
    print(f"   {n:42s} median {np.median(seg):8.0f}  by wave {np.median(seg, axis=(0, 1)).astype(int)}")
