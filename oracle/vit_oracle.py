"""CPU restatement of the reference ViT forward (TEST INFRASTRUCTURE, not product code).

This file is the parity oracle for the MI355X forward path.  It restates, in plain
tensor arithmetic on the CPU, what the reference's `vitm.models` modules compute, one
function per reference function, each citing the reference file:line it follows.  It is
imported only by the tests, by ``__graft_entry__.smoke()`` and by ``bench.py``'s
``cpu_baseline`` leg -- never by the product package, which must fail loudly when its
HIP library is missing.

Pinning: the reference's own tests are shape-only (no numeric golden vectors), so this
oracle is pinned by fixtures generated in the build container from the imported
reference modules (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and checked
in ``tests/test_oracle_golden.py``.

All functions take a flat ``state_dict`` with the reference's key names (as produced by
``VisionTransformer.state_dict()`` or, with the ``vit_model.`` prefix and ``head.*``,
by ``VITC.state_dict()``) and a small ``cfg`` dict:

    cfg = dict(image_size=(H, W), patch_size=(p, p), num_heads=H, depth=L)

Arithmetic is carried out in ``dtype`` (float32 mirrors the reference; float64 gives a
high-precision truth for tolerance budgeting).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

LN_EPS = 1e-5  # nn.LayerNorm default; the reference never overrides it (models.py:189,196,310,324)


def _t(sd: Dict[str, torch.Tensor], key: str, dtype) -> Optional[torch.Tensor]:
    v = sd.get(key)
    return None if v is None else v.detach().to(device="cpu", dtype=dtype)


def to_2tuple(x):
    """utils/helper.py:14-21 -- repeat a scalar twice, pass iterables through."""
    if isinstance(x, (tuple, list)):
        return tuple(x)
    return (x, x)


def layer_norm(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor],
               eps: float = LN_EPS) -> torch.Tensor:
    """nn.LayerNorm over the last dim: biased variance, eps inside the sqrt, affine."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    y = xc / torch.sqrt(var + eps)
    if weight is not None:
        y = y * weight
    if bias is not None:
        y = y + bias
    return y


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() with approximate='none' (models.py:106,122): 0.5 x (1 + erf(x / sqrt 2))."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def im2col(img: torch.Tensor, patch: Tuple[int, int]) -> torch.Tensor:
    """(B,C,H,W) -> (B, gh*gw, C*ph*pw); row order (py,px), column order (c,ky,kx).

    That column order is the one ``conv.weight.reshape(D, -1)`` has, so the strided
    Conv2d(k=stride=patch) of patch_embeddings.py:67,115 equals ``im2col(x) @ W.reshape(D,-1).T``.
    """
    B, C, H, W = img.shape
    ph, pw = patch
    gh, gw = H // ph, W // pw
    x = img[:, :, : gh * ph, : gw * pw].reshape(B, C, gh, ph, gw, pw)
    x = x.permute(0, 2, 4, 1, 3, 5)  # B, gh, gw, C, ph, pw
    return x.reshape(B, gh * gw, C * ph * pw)


def patch_embed(img: torch.Tensor, sd, cfg, prefix: str = "patch_embed.", dtype=torch.float32) -> torch.Tensor:
    """PatchEmbed.forward, flatten=True branch (patch_embeddings.py:111-121)."""
    ih, iw = to_2tuple(cfg["image_size"])
    B, C, H, W = img.shape
    assert ih == H, "Input tensor's height does not match the model"   # patch_embeddings.py:113
    assert iw == W, "Input tensor's width does not match the model"    # patch_embeddings.py:114
    w = _t(sd, prefix + "projection.weight", dtype)                   # (D, C, ph, pw)
    b = _t(sd, prefix + "projection.bias", dtype)
    patch = (w.shape[2], w.shape[3])
    cols = im2col(img.to(dtype), patch)
    tok = cols @ w.reshape(w.shape[0], -1).t()                        # conv == GEMM (:115) + flatten/transpose (:117)
    if b is not None:
        tok = tok + b
    nw = _t(sd, prefix + "norm_layer.weight", dtype)
    if nw is not None:                                                # optional norm (:120)
        tok = layer_norm(tok, nw, _t(sd, prefix + "norm_layer.bias", dtype))
    return tok


def forward_features(img: torch.Tensor, sd, cfg, prefix: str = "", dtype=torch.float32) -> torch.Tensor:
    """VisionTransformer.forward_features (models.py:352-361).

    cat(cls, tokens) -> norm_pre -> + pos_embed.  Note norm_pre comes BEFORE the
    positional add (models.py:359-360).
    """
    x = patch_embed(img, sd, cfg, prefix + "patch_embed.", dtype)
    B = x.shape[0]
    cls = _t(sd, prefix + "class_token", dtype)
    if cls is not None:
        x = torch.cat((cls.expand(B, -1, -1), x), dim=1)              # models.py:356-358
    nw = _t(sd, prefix + "norm_pre.weight", dtype)
    if nw is not None:
        x = layer_norm(x, nw, _t(sd, prefix + "norm_pre.bias", dtype))  # models.py:359
    pos = _t(sd, prefix + "pos_embed", dtype)
    if pos is not None:
        x = x + pos                                                   # models.py:360
    return x


def attention(x: torch.Tensor, sd, num_heads: int, prefix: str, dtype=torch.float32) -> torch.Tensor:
    """Attention.forward, unfused branch (models.py:72-95)."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = x @ _t(sd, prefix + "qkv.weight", dtype).t()                # models.py:76
    qb = _t(sd, prefix + "qkv.bias", dtype)
    if qb is not None:
        qkv = qkv + qb
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)  # models.py:79
    q, k, v = qkv[0], qkv[1], qkv[2]                                  # models.py:81
    qn = _t(sd, prefix + "q_norm.weight", dtype)
    if qn is not None:                                                # qk_norm=True (models.py:54-55,82)
        q = layer_norm(q, qn, _t(sd, prefix + "q_norm.bias", dtype))
        k = layer_norm(k, _t(sd, prefix + "k_norm.weight", dtype), _t(sd, prefix + "k_norm.bias", dtype))
    q = q * (hd ** -0.5)                                              # models.py:88
    s = q @ k.transpose(-2, -1)                                       # models.py:89
    s = s - s.amax(dim=-1, keepdim=True)
    p = torch.exp(s)
    p = p / p.sum(dim=-1, keepdim=True)                               # softmax(-1), models.py:90
    o = p @ v                                                         # models.py:91
    o = o.transpose(1, 2).reshape(B, N, C)                            # models.py:93
    o = o @ _t(sd, prefix + "proj.weight", dtype).t()                 # models.py:94
    pb = _t(sd, prefix + "proj.bias", dtype)
    if pb is not None:
        o = o + pb
    return o


def parallel_block(x: torch.Tensor, sd, num_heads: int, prefix: str = "", dtype=torch.float32) -> torch.Tensor:
    """ParallelBlock.forward (parallel_block.py:193-221): LayerNorm -> one projection to [q | k | v | mlp hidden]
    -> attention with per-head LayerNorm of q and k (scores scaled AFTER q.k^T, parallel_block.py:88-91) in parallel
    with GELU + out_proj (:121-124) -> concat -> fused_proj_2 -> LayerScale -> + x."""
    B, N, C = x.shape
    hd = C // num_heads
    nx = layer_norm(x, _t(sd, prefix + "norm.weight", dtype), _t(sd, prefix + "norm.bias", dtype))
    f1 = nx @ _t(sd, prefix + "fused_proj_1.weight", dtype).t() + _t(sd, prefix + "fused_proj_1.bias", dtype)
    qkv, hid = f1[..., :3 * C], f1[..., 3 * C:]
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = layer_norm(q, _t(sd, prefix + "attn.q_norm.weight", dtype), _t(sd, prefix + "attn.q_norm.bias", dtype))
    k = layer_norm(k, _t(sd, prefix + "attn.k_norm.weight", dtype), _t(sd, prefix + "attn.k_norm.bias", dtype))
    s = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    s = s - s.amax(dim=-1, keepdim=True)
    p = torch.exp(s)
    p = p / p.sum(dim=-1, keepdim=True)
    a = (p @ v).permute(0, 2, 1, 3).reshape(B, N, C)
    a = a @ _t(sd, prefix + "attn.proj.weight", dtype).t() + _t(sd, prefix + "attn.proj.bias", dtype)
    m = gelu_erf(hid) @ _t(sd, prefix + "mlp.out_proj.weight", dtype).t()
    mb = _t(sd, prefix + "mlp.out_proj.bias", dtype)
    if mb is not None:
        m = m + mb
    f2 = torch.cat((a, m), dim=-1) @ _t(sd, prefix + "fused_proj_2.weight", dtype).t() + _t(sd, prefix + "fused_proj_2.bias", dtype)
    return x + f2 * _t(sd, prefix + "layer_scale.gamma", dtype)


def mlp(x: torch.Tensor, sd, prefix: str, dtype=torch.float32) -> torch.Tensor:
    """MLP.forward (models.py:126-130): linear_layer1 -> GELU(erf) -> linear_layer2."""
    h = x @ _t(sd, prefix + "linear_layer1.weight", dtype).t()
    b1 = _t(sd, prefix + "linear_layer1.bias", dtype)
    if b1 is not None:
        h = h + b1
    h = gelu_erf(h)
    y = h @ _t(sd, prefix + "linear_layer2.weight", dtype).t()
    b2 = _t(sd, prefix + "linear_layer2.bias", dtype)
    if b2 is not None:
        y = y + b2
    return y


def block(x: torch.Tensor, sd, num_heads: int, prefix: str, dtype=torch.float32) -> torch.Tensor:
    """Block.forward (models.py:208-211) with LayerScale (models.py:138-139)."""
    h = layer_norm(x, _t(sd, prefix + "norm1.weight", dtype), _t(sd, prefix + "norm1.bias", dtype))
    x = x + attention(h, sd, num_heads, prefix + "attn.", dtype) * _t(sd, prefix + "ls1.gamma", dtype)
    h = layer_norm(x, _t(sd, prefix + "norm2.weight", dtype), _t(sd, prefix + "norm2.bias", dtype))
    x = x + mlp(h, sd, prefix + "mlp.", dtype) * _t(sd, prefix + "ls2.gamma", dtype)
    return x


def _depth(sd, prefix: str) -> int:
    n = 0
    while (prefix + f"blocks.{n}.norm1.weight") in sd:
        n += 1
    return n


def forward_heads(x: torch.Tensor, sd, cfg, prefix: str = "", dtype=torch.float32, stages=None) -> torch.Tensor:
    """VisionTransformer.forward_heads (models.py:408-411), with the intended sequential
    meaning of ``self.blocks(x)`` (the ModuleList is iterated in order)."""
    for i in range(_depth(sd, prefix)):
        x = block(x, sd, cfg["num_heads"], prefix + f"blocks.{i}.", dtype)
        if stages is not None:
            stages.append(x)
    nw = _t(sd, prefix + "norm_final.weight", dtype)
    if nw is not None:
        x = layer_norm(x, nw, _t(sd, prefix + "norm_final.bias", dtype))
    return x


def vit_forward(img: torch.Tensor, sd, cfg, prefix: str = "", dtype=torch.float32, stages=None) -> torch.Tensor:
    """VisionTransformer.forward (models.py:413-416): (B,C,H,W) -> (B,N,D)."""
    x = forward_features(img, sd, cfg, prefix, dtype)
    if stages is not None:
        stages.append(x)
    return forward_heads(x, sd, cfg, prefix, dtype, stages)


def vitc_forward(img: torch.Tensor, sd, cfg, dtype=torch.float32, return_embeddings: bool = False):
    """VITC.forward (classifier.py:35-39): head(vit(x)[:, 0])."""
    emb = vit_forward(img, sd, cfg, "vit_model.", dtype)
    logits = emb[:, 0] @ _t(sd, "head.weight", dtype).t() + _t(sd, "head.bias", dtype)
    if return_embeddings:
        return logits, emb
    return logits


def flops_per_image(cfg_shapes: dict) -> float:
    """Algorithmic FLOPs (MAC = 2) of one classification forward; LN/softmax/GELU excluded.

    2(N-1) K_pe D + L (2 N D 3D + 4 N^2 D + 2 N D^2 + 4 N D hid) + 2 D C
    """
    N, D, L = cfg_shapes["tokens"], cfg_shapes["embed_dim"], cfg_shapes["depth"]
    hid, C, kpe = cfg_shapes["hidden"], cfg_shapes.get("num_classes", 0), cfg_shapes["k_pe"]
    return (2.0 * (N - 1) * kpe * D
            + L * (2.0 * N * D * 3 * D + 4.0 * N * N * D + 2.0 * N * D * D + 4.0 * N * D * hid)
            + 2.0 * D * C)
