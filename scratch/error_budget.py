"""Where the bf16 path's logit error comes from: the fp32 forward on the CPU with ONE kind of tensor rounded to bf16
at a time (the places where the HIP bf16 path stores a bf16 tensor), then all of them together.

    python scratch/error_budget.py [config ...]        (CPU only; tiny2 and vit_tiny_cifar by default)

Rounding points: `w` packed weights (with the LayerNorm gamma folded in, as vitm_prepack_linear stores them), `img`
the input image, `x` the residual stream after every update, `qkv` the projection output, `p` the softmax
probabilities fed to P.V, `o` the attention output, `h` the GELU output.  Everything else stays fp32, as on the GPU
(accumulators, LayerNorm statistics, softmax sums, epilogue arithmetic).
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import weights as W  # noqa: E402


def forward(x, sd, cfg, on):
    def r(name, t):
        return t.bfloat16().float() if name in on else t

    def t(k):
        return torch.from_numpy(sd[k])

    def ln_stats(v):
        mu = v.mean(-1, keepdim=True)
        return mu, torch.rsqrt(((v - mu) ** 2).mean(-1, keepdim=True) + 1e-5)

    def ln(v, pre):
        mu, rs = ln_stats(v)
        return (v - mu) * rs * t(pre + "weight") + t(pre + "bias")

    def folded(v, pre_ln, wkey, bkey=None):
        # LN(v) W^T + b computed the way the GEMM does: raw v against W' = W diag(gamma) (rounded), statistics applied after
        w, g, be = t(wkey), t(pre_ln + "weight"), t(pre_ln + "bias")
        wp = r("w", w * g)
        mu, rs = ln_stats(v)
        y = rs * (v @ wp.t() - mu * wp.sum(1)) + be @ w.t()
        return y + t(bkey) if bkey else y

    v = "vit_model."
    p = cfg["patch_size"]
    B = x.shape[0]
    g = cfg["image_size"] // p
    cols = r("img", x).reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)
    wpe = t(v + "patch_embed.projection.weight")
    tok = r("x", cols @ r("w", wpe.reshape(wpe.shape[0], -1)).t())
    if cfg["embed_norm"]:
        tok = ln(tok, v + "patch_embed.norm_layer.")
    xs = torch.cat((t(v + "class_token").expand(B, -1, -1), tok), 1)
    if cfg["norm_pre"]:
        xs = ln(xs, v + "norm_pre.")
    xs = r("x", xs + t(v + "pos_embed"))
    H = cfg["num_heads"]
    D = cfg["embed_dim"]
    hd = D // H
    N = xs.shape[1]
    for i in range(cfg["depth"]):
        b = v + f"blocks.{i}."
        qkv = r("qkv", folded(xs, b + "norm1.", b + "attn.qkv.weight")).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        s = (qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1)
        e = torch.exp(s - s.amax(-1, keepdim=True))
        o = (r("p", e) @ qkv[2]) / e.sum(-1, keepdim=True)
        o = r("o", o.transpose(1, 2).reshape(B, N, D))
        xs = r("x", xs + (o @ r("w", t(b + "attn.proj.weight")).t() + t(b + "attn.proj.bias")) * t(b + "ls1.gamma"))
        h = folded(xs, b + "norm2.", b + "mlp.linear_layer1.weight", b + "mlp.linear_layer1.bias")
        h = r("h", 0.5 * h * (1 + torch.erf(h / math.sqrt(2))))
        xs = r("x", xs + (h @ r("w", t(b + "mlp.linear_layer2.weight")).t() + t(b + "mlp.linear_layer2.bias")) * t(b + "ls2.gamma"))
    cls = xs[:, 0]
    if cfg["final_norm"]:
        return folded(cls, v + "norm_final.", "head.weight", "head.bias")
    return cls @ r("w", t("head.weight")).t() + t("head.bias")


def main(names):
    torch.set_grad_enabled(False)
    for name in names:
        cfg = W.CONFIGS[name]
        batch = 64 if cfg["image_size"] <= 32 else 2
        sd = W.make_state_dict(cfg, 0)
        x = torch.from_numpy(W.make_input(cfg, batch, 1))
        ref = forward(x, sd, cfg, ())
        print(f"{name}: {ref.numel()} logits, max|logit| {ref.abs().max():.2f}")
        rows = [(k,) for k in ("w", "img", "x", "qkv", "p", "o", "h")] + [("w", "img", "x", "qkv", "p", "o", "h")]
        for on in rows:
            d = forward(x, sd, cfg, on) - ref
            print(f"  bf16 {'+'.join(on):22s} max {d.abs().max():.4f}  rms {d.pow(2).mean().sqrt():.4f}")


if __name__ == "__main__":
    main(sys.argv[1:] or ["tiny2", "vit_tiny_cifar"])
