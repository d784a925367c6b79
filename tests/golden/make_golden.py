"""Generate the golden fixtures from the IMPORTED reference (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports ``vitm.models`` from /root/reference, loads the deterministic synthetic weights of
``tests/golden/weights.py`` with ``load_state_dict(strict=True)`` (which also proves the key
names/shapes are the reference's), runs the reference modules in fp32 on the CPU and stores
inputs' seeds and the reference's outputs as small ``.npz`` files next to this script.

``VisionTransformer.forward`` cannot be called at the reference's HEAD (``self.blocks`` is an
``nn.ModuleList`` that is *called*, models.py:312,409), so the blocks are iterated in order,
which is the only meaning the code can have: forward_features -> for blk in blocks -> norm_final.
``embed_dim`` is not forwarded to PatchEmbed (models.py:291-295), hence the ``functools.partial``.
"""
from __future__ import annotations

import contextlib
import functools
import io
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import weights as W  # noqa: E402

from vitm.models import PatchEmbed, VisionTransformer, VITC  # noqa: E402


def build_reference(cfg: dict, depth=None):
    LN = nn.LayerNorm
    vt = VisionTransformer(
        cfg["image_size"], patch_size=cfg["patch_size"], num_classes=cfg["num_classes"],
        num_heads=cfg["num_heads"], mlp_ratio=cfg["mlp_ratio"],
        depth=cfg["depth"] if depth is None else depth, embed_dim=cfg["embed_dim"],
        norm_layer=LN,
        embed_norm_layer=LN if cfg["embed_norm"] else None,
        final_norm_layer=LN if cfg["final_norm"] else None,
        embed_layer=functools.partial(PatchEmbed, embed_dim=cfg["embed_dim"], bias=cfg["patch_bias"]),
    )
    return VITC(vt).eval()


@torch.no_grad()
def run_reference(model, x: torch.Tensor):
    vt = model.vit_model
    with contextlib.redirect_stdout(io.StringIO()):      # PatchEmbed.forward prints (patch_embeddings.py:112)
        f = vt.forward_features(x)
    stages = [f]
    for blk in vt.blocks:
        f = blk(f)
        stages.append(f)
    emb = vt.norm_final(f)
    logits = model.head(emb[:, 0])
    return stages, emb, logits


def gen(name: str, batch: int, seed: int, depth=None, keep_stages=True, emb_rows=None, tag=None):
    """``tag``: file name stem when it differs from the configuration name (a cut-depth variant)."""
    cfg = W.CONFIGS[name]
    sd = W.make_state_dict(cfg, seed, depth)
    model = build_reference(cfg, depth)
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    x = torch.from_numpy(W.make_input(cfg, batch, seed + 1))
    stages, emb, logits = run_reference(model, x)
    out = dict(logits=logits.numpy(), batch=np.int64(batch), seed=np.int64(seed),
               depth=np.int64(cfg["depth"] if depth is None else depth))
    # the reference's OWN bf16 forward (module.bfloat16(), bf16 input) on the same weights: how far a bf16
    # forward of this network lands from the fp32 one, as a yardstick for the HIP bf16 path
    _, _, logits_bf16 = run_reference(model.bfloat16(), x.bfloat16())
    out["logits_ref_bf16"] = logits_bf16.float().numpy()
    if emb_rows is None:
        out["embeddings"] = emb.numpy()
    else:  # large configs: keep the cls row and a strided sample of token rows
        out["embeddings_rows"] = np.asarray(emb_rows, dtype=np.int64)
        out["embeddings"] = emb[:, emb_rows].numpy()
    if keep_stages:
        out["features"] = stages[0].numpy()
        for i, s in enumerate(stages[1:]):
            out[f"block{i}"] = s.numpy()
    path = os.path.join(HERE, f"{tag or name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), "
          f"max|logit|={logits.abs().max():.3f}, reference bf16-vs-fp32 logit err "
          f"{(logits_bf16.float() - logits).abs().max():.4f}")


if __name__ == "__main__":
    torch.manual_seed(0)
    gen("tiny2", batch=2, seed=0)
    gen("cifar_recipe", batch=1, seed=3)
    gen("vit_tiny_cifar", batch=4, seed=5, keep_stages=False)
    gen("vit_b16", batch=2, seed=7, keep_stages=False, emb_rows=[0, 1, 50, 99, 196])
    gen("vit_s16", batch=2, seed=9, keep_stages=False, emb_rows=[0, 1, 50, 99, 196])
    gen("vit_b16_384", batch=1, seed=11, keep_stages=False, emb_rows=[0, 1, 100, 300, 576])
