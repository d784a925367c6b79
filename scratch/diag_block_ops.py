"""Each launch of block 0 on one shard against torch fp32 arithmetic on that launch's own inputs.
python scratch/diag_block_ops.py vit_s16 256 32 160"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vision-transformer_amd"), os.path.join(ROOT, "tests"),
                os.path.join(ROOT, "tests", "golden")]
import weights as W                      # noqa: E402
from test_gpu_model import build, GOLD   # noqa: E402
from vitm_amd import _ops                # noqa: E402
from vitm_amd.models.patch_embeddings import layernorm_params   # noqa: E402

name, batch, shard, img = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = np.load(os.path.join(GOLD, f"{name}.npz"))
seed, gb = int(g["seed"]), int(g["batch"])
model, sd, cfg = build(name, seed)
xg = torch.from_numpy(W.make_input(cfg, gb, seed + 1))
rest = torch.from_numpy(W.make_input(cfg, batch - gb, seed + 2))
x = torch.cat([xg, rest]).cuda().to(torch.bfloat16)
s0 = img // shard * shard
vt = model.vit_model
blk = vt.blocks[0]
attn, mlp = blk.attn, blk.mlp


def report(what, got, ref):
    got, ref = got.float(), ref.float()
    d = (got - ref).abs()
    rows = d.amax(dim=1)
    top = rows.topk(5)
    print(f"{what:10s} scale {ref.abs().max().item():8.3f} max err {d.max().item():.5f} rms {d.pow(2).mean().sqrt().item():.6f} "
          f"worst rows {top.indices.tolist()} {[round(v, 4) for v in top.values.tolist()]}")


with torch.no_grad():
    h = vt.forward_features(x[s0:s0 + shard])
    B, N, D = h.shape
    M = B * N
    x2d = h.reshape(M, D).clone()
    ws = blk._workspace(M, x.device, torch.bfloat16)
    ln1 = layernorm_params(blk.norm1, "n1")
    ln2 = layernorm_params(blk.norm2, "n2")
    dt = torch.bfloat16
    for pk, nm in ((attn._qkv_pack(dt, ln1), "qkv"), (attn._proj_pack(dt, blk.ls1.gamma), "proj"),
                   (mlp._fc1_pack(dt, ln2), "fc1"), (mlp._fc2_pack(dt, blk.ls2.gamma), "fc2")):
        print(nm, "big tile" if _ops.uses_big_tile(M, pk) else "128x128", "N", pk.N, "K", pk.K)
    # 1. statistics
    _ops.ln_stats(x2d, M, D, ln1[2], ws["stats"])
    xf = x2d.float()
    mean = xf.mean(1)
    rstd = (xf.var(1, unbiased=False) + ln1[2]).rsqrt()
    report("ln_stats", ws["stats"], torch.stack([mean, rstd], 1))
    # 2. qkv
    _ops.gemm(x2d, attn._qkv_pack(dt, ln1), ws["qkv"], _ops.EPI_LN, M, stats=ws["stats"])
    ref = F.linear(F.layer_norm(xf, (D,), ln1[0].float(), ln1[1].float(), ln1[2]), attn.qkv.weight.float())
    report("qkv", ws["qkv"], ref)
    # 3. attention
    attn._core(ws["qkv"], ws["o"], B, N)
    q, k, v = ws["qkv"].float().reshape(B, N, 3, attn.num_heads, attn.head_dim).permute(2, 0, 3, 1, 4)
    o = torch.softmax((q * attn.scale) @ k.transpose(-1, -2), -1) @ v
    report("attention", ws["o"][:, :D], o.transpose(1, 2).reshape(M, D))
    # 4. proj + LayerScale + residual
    ref = xf + blk.ls1.gamma.float() * F.linear(ws["o"][:, :D].float(), attn.proj.weight.float(), attn.proj.bias.float())
    _ops.gemm(ws["o"], attn._proj_pack(dt, blk.ls1.gamma), x2d, _ops.EPI_RESIDUAL, M, resid=x2d)
    report("proj", x2d, ref)
    # 5. fc1
    xf = x2d.float()
    _ops.ln_stats(x2d, M, D, ln2[2], ws["stats"])
    mean = xf.mean(1)
    rstd = (xf.var(1, unbiased=False) + ln2[2]).rsqrt()
    report("ln_stats 2", ws["stats"], torch.stack([mean, rstd], 1))
    mlp._fc1(x2d, mlp._fc1_pack(dt, ln2), ws["h"], M, ln_epilogue=True, stats=ws["stats"])
    ref = F.gelu(F.linear(F.layer_norm(xf, (D,), ln2[0].float(), ln2[1].float(), ln2[2]), mlp.linear_layer1.weight.float(),
                          mlp.linear_layer1.bias.float()))
    report("fc1", ws["h"][:, :ref.shape[1]], ref)
    # 6. fc2
    ref = xf + blk.ls2.gamma.float() * F.linear(ws["h"][:, :mlp.hidden_features].float(), mlp.linear_layer2.weight.float(),
                                                mlp.linear_layer2.bias.float())
    _ops.gemm(ws["h"], mlp._fc2_pack(dt, blk.ls2.gamma), x2d, _ops.EPI_RESIDUAL, M, resid=x2d)
    report("fc2", x2d, ref)
