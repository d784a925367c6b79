"""Block-by-block error of the two bf16 GEMM routes (whole batch / one shard) against the fp32 forward, per image.
python scratch/diag_stages.py vit_s16 256 32 160"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-transformer_amd"), os.path.join(ROOT, "tests"),
                os.path.join(ROOT, "tests", "golden")]
import weights as W                      # noqa: E402
from test_gpu_model import build, GOLD   # noqa: E402

name, batch, shard, img = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = np.load(os.path.join(GOLD, f"{name}.npz"))
seed, gb = int(g["seed"]), int(g["batch"])
model, sd, cfg = build(name, seed)
m32, _, _ = build(name, seed, dtype=torch.float32)
xg = torch.from_numpy(W.make_input(cfg, gb, seed + 1))
rest = torch.from_numpy(W.make_input(cfg, batch - gb, seed + 2))
x32 = torch.cat([xg, rest]).cuda()
x = x32.to(torch.bfloat16)
s0 = img // shard * shard
sl = slice(s0, s0 + shard)


def stages(m, xin):
    vt = m.vit_model
    out = []
    with torch.no_grad():
        h = vt.forward_features(xin)
        out.append(h.float())
        for blk in vt.blocks:
            h = blk(h)
            out.append(h.float())
    return out


full = [s[sl] for s in stages(model, x)]
sh = stages(model, x[sl])
ref = stages(m32, x32[sl])
k = img - s0
for i, (f, s, r) in enumerate(zip(full, sh, ref)):
    scale = r.abs().amax(dim=(1, 2))
    ef = (f - r).abs().amax(dim=(1, 2)) / scale
    es = (s - r).abs().amax(dim=(1, 2)) / scale
    rf = (f - r).pow(2).mean(dim=(1, 2)).sqrt() / r.pow(2).mean(dim=(1, 2)).sqrt()
    rs = (s - r).pow(2).mean(dim=(1, 2)).sqrt() / r.pow(2).mean(dim=(1, 2)).sqrt()
    # token rows of the shard run that differ most from the whole-batch run
    drow = (s[k] - f[k]).abs().amax(dim=1)
    top = drow.topk(3)
    print(f"stage {i:2d} image {img}: rel-rms full {rf[k]:.5f} shard {rs[k]:.5f} | median over shard: full {rf.median():.5f} "
          f"shard {rs.median():.5f} max shard {rs.max():.5f} (image {s0 + rs.argmax().item()}) | worst rows {top.indices.tolist()} "
          f"{[round(v, 4) for v in top.values.tolist()]} scale {scale[k]:.2f}")
