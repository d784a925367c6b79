"""Where do the whole-batch (256x256 tiles) and sharded (128x128 tiles) bf16 forwards of one network differ most,
and which of the two is further from the fp32 forward on those images?   python scratch/diag_full_size.py vit_s16 256 32"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-transformer_amd"), os.path.join(ROOT, "tests"),
                os.path.join(ROOT, "tests", "golden")]
import weights as W                      # noqa: E402
from test_gpu_model import build, GOLD   # noqa: E402

name, batch, shard = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
fixture = sys.argv[4] if len(sys.argv) > 4 else name
depth = int(sys.argv[5]) if len(sys.argv) > 5 else None
g = np.load(os.path.join(GOLD, f"{fixture}.npz"))
seed, gb = int(g["seed"]), int(g["batch"])
model, sd, cfg = build(name, seed, depth=depth)
m32, _, _ = build(name, seed, depth=depth, dtype=torch.float32)
xg = torch.from_numpy(W.make_input(cfg, gb, seed + 1))
rest = torch.from_numpy(W.make_input(cfg, batch - gb, seed + 2))
x32 = torch.cat([xg, rest]).cuda()
x = x32.to(torch.bfloat16)
with torch.no_grad():
    full = model(x).float()
    shards = torch.cat([model(x[i:i + shard]) for i in range(0, batch, shard)]).float()
    ref = torch.cat([m32(x32[i:i + 32]) for i in range(0, batch, 32)])
d = (full - shards).abs()
ef, es = (full - ref).abs(), (shards - ref).abs()
print(f"|logit| max {ref.abs().max().item():.3f}  rms {ref.pow(2).mean().sqrt().item():.3f}")
print(f"full vs shards: max {d.max().item():.4f} rms {d.pow(2).mean().sqrt().item():.4f}")
print(f"full vs fp32  : max {ef.max().item():.4f} rms {ef.pow(2).mean().sqrt().item():.4f}")
print(f"shards vs fp32: max {es.max().item():.4f} rms {es.pow(2).mean().sqrt().item():.4f}")
per_img = d.max(dim=1).values
top = per_img.topk(8)
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    j = d[i].argmax().item()
    print(f" image {i:4d} col {j:4d}: full {full[i, j]:.4f} shard {shards[i, j]:.4f} fp32 {ref[i, j]:.4f}  "
          f"img rms full-fp32 {ef[i].pow(2).mean().sqrt().item():.4f} shard-fp32 {es[i].pow(2).mean().sqrt().item():.4f} "
          f"|logit|max {ref[i].abs().max().item():.2f}")
q = torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999, 1.0], device=d.device)
print("quantiles of |full-shards|:", torch.quantile(d.flatten()[:4_000_000], q).tolist())
print("per-image rms(full-fp32) quantiles:", torch.quantile(ef.pow(2).mean(1).sqrt(), q).tolist())
