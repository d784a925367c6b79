"""Deterministic synthetic weights and inputs shared by the golden generator and the tests.

Weights come from ``numpy.random.RandomState`` (the frozen legacy MT19937 stream), so the
same ``(config, seed)`` reproduces bit-identical fp32 parameters on any machine without
storing them.  Key names and shapes are exactly those of the reference's
``VITC.state_dict()`` (``vit_model.*`` + ``head.*``).

The distributions are "trained-like" rather than the reference's init: LayerScale gamma is
~N(1, 0.1) instead of 1e-5 (at 1e-5 every block is numerically invisible), LayerNorm
affines and all biases are non-trivial, and the qkv projection has gain 2 so the softmax
is peaked.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

CONFIGS: Dict[str, dict] = {
    # small all-features config: patch norm + norm_pre + final norm, 2 heads of 64
    "tiny2": dict(image_size=32, patch_size=4, embed_dim=128, num_heads=2, depth=2, mlp_ratio=4.0,
                  num_classes=10, embed_norm=True, norm_pre=True, final_norm=True, patch_bias=False),
    # benchmark configuration 1: ViT-tiny on CIFAR-10 32x32, patch 4
    "vit_tiny_cifar": dict(image_size=32, patch_size=4, embed_dim=192, num_heads=3, depth=12, mlp_ratio=4.0,
                           num_classes=10, embed_norm=False, norm_pre=True, final_norm=True, patch_bias=False),
    # the reference's own CIFAR recipe (training/train.py:167-175): head_dim 96, hidden int(768*0.8)=614
    "cifar_recipe": dict(image_size=32, patch_size=4, embed_dim=768, num_heads=8, depth=1, mlp_ratio=0.8,
                         num_classes=10, embed_norm=True, norm_pre=True, final_norm=True, patch_bias=False),
    # the same recipe at the depth the reference trains (VisionTransformer's default depth 12): the workload of
    # `python -m vitm.training.train` itself; bench-only (no fixture)
    "cifar_recipe_full": dict(image_size=32, patch_size=4, embed_dim=768, num_heads=8, depth=12, mlp_ratio=0.8,
                              num_classes=10, embed_norm=True, norm_pre=True, final_norm=True, patch_bias=False),
    # benchmark configuration 2: ViT-S/16 224 (embedding forward)
    "vit_s16": dict(image_size=224, patch_size=16, embed_dim=384, num_heads=6, depth=12, mlp_ratio=4.0,
                    num_classes=1000, embed_norm=False, norm_pre=True, final_norm=True, patch_bias=False),
    # benchmark configuration 3: ViT-B/16 224 (classification forward)
    "vit_b16": dict(image_size=224, patch_size=16, embed_dim=768, num_heads=12, depth=12, mlp_ratio=4.0,
                    num_classes=1000, embed_norm=False, norm_pre=True, final_norm=True, patch_bias=False),
    # benchmark configuration 4: ViT-B/16 384 (577 tokens)
    "vit_b16_384": dict(image_size=384, patch_size=16, embed_dim=768, num_heads=12, depth=12, mlp_ratio=4.0,
                        num_classes=1000, embed_norm=False, norm_pre=True, final_norm=True, patch_bias=False),
    # benchmark configuration 5 (forward part): ViT-L/16 224
    "vit_l16": dict(image_size=224, patch_size=16, embed_dim=1024, num_heads=16, depth=24, mlp_ratio=4.0,
                    num_classes=1000, embed_norm=False, norm_pre=True, final_norm=True, patch_bias=False),
}


def shapes(cfg: dict) -> dict:
    img, p, D = cfg["image_size"], cfg["patch_size"], cfg["embed_dim"]
    g = img // p
    return dict(tokens=g * g + 1, embed_dim=D, depth=cfg["depth"], hidden=int(D * cfg["mlp_ratio"]),
                num_classes=cfg["num_classes"], k_pe=3 * p * p, heads=cfg["num_heads"])


def _xavier(rs, out_f, in_f, gain=1.0):
    a = gain * math.sqrt(6.0 / (in_f + out_f))
    return rs.uniform(-a, a, size=(out_f, in_f)).astype(np.float32)


def _ln(rs, dim, sd, prefix):
    sd[prefix + "weight"] = (1.0 + 0.1 * rs.standard_normal(dim)).astype(np.float32)
    sd[prefix + "bias"] = (0.1 * rs.standard_normal(dim)).astype(np.float32)


def make_state_dict(cfg: dict, seed: int = 0, depth: int | None = None) -> Dict[str, np.ndarray]:
    """fp32 parameters under the reference's VITC key names."""
    rs = np.random.RandomState(seed)
    D, p, C = cfg["embed_dim"], cfg["patch_size"], cfg["num_classes"]
    L = cfg["depth"] if depth is None else depth
    hid = int(D * cfg["mlp_ratio"])
    s = shapes(cfg)
    sd: Dict[str, np.ndarray] = {}
    v = "vit_model."
    kpe = 3 * p * p
    sd[v + "patch_embed.projection.weight"] = (
        rs.standard_normal((D, 3, p, p)) * math.sqrt(2.0 / kpe)).astype(np.float32)
    if cfg.get("patch_bias"):
        sd[v + "patch_embed.projection.bias"] = (0.02 * rs.standard_normal(D)).astype(np.float32)
    if cfg.get("embed_norm"):
        _ln(rs, D, sd, v + "patch_embed.norm_layer.")
    sd[v + "class_token"] = (0.5 * rs.standard_normal((1, 1, D))).astype(np.float32)
    sd[v + "pos_embed"] = (0.2 * rs.standard_normal((1, s["tokens"], D))).astype(np.float32)
    if cfg.get("norm_pre"):
        _ln(rs, D, sd, v + "norm_pre.")
    for i in range(L):
        b = v + f"blocks.{i}."
        _ln(rs, D, sd, b + "norm1.")
        sd[b + "attn.qkv.weight"] = _xavier(rs, 3 * D, D, gain=2.0)
        sd[b + "attn.proj.weight"] = _xavier(rs, D, D)
        sd[b + "attn.proj.bias"] = (0.02 * rs.standard_normal(D)).astype(np.float32)
        sd[b + "ls1.gamma"] = (1.0 + 0.1 * rs.standard_normal(D)).astype(np.float32)
        _ln(rs, D, sd, b + "norm2.")
        sd[b + "mlp.linear_layer1.weight"] = _xavier(rs, hid, D)
        sd[b + "mlp.linear_layer1.bias"] = (0.02 * rs.standard_normal(hid)).astype(np.float32)
        sd[b + "mlp.linear_layer2.weight"] = _xavier(rs, D, hid)
        sd[b + "mlp.linear_layer2.bias"] = (0.02 * rs.standard_normal(D)).astype(np.float32)
        sd[b + "ls2.gamma"] = (1.0 + 0.1 * rs.standard_normal(D)).astype(np.float32)
    if cfg.get("final_norm"):
        _ln(rs, D, sd, v + "norm_final.")
    sd["head.weight"] = _xavier(rs, C, D)
    sd["head.bias"] = (0.02 * rs.standard_normal(C)).astype(np.float32)
    return sd


def make_input(cfg: dict, batch: int, seed: int = 1) -> np.ndarray:
    """U(-1,1) NCHW fp32 image batch (the range ToTensor+Normalize(0.5,0.5) produces)."""
    rs = np.random.RandomState(seed)
    img = cfg["image_size"]
    return rs.uniform(-1.0, 1.0, size=(batch, 3, img, img)).astype(np.float32)


def oracle_cfg(cfg: dict) -> dict:
    """The small dict ``oracle.vit_oracle`` functions take."""
    return dict(image_size=(cfg["image_size"], cfg["image_size"]),
                patch_size=(cfg["patch_size"], cfg["patch_size"]),
                num_heads=cfg["num_heads"], depth=cfg["depth"])
