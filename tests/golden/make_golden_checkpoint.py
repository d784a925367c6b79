"""Checkpoint fixtures in the reference's two on-disk formats, written from the IMPORTED reference modules
(build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_checkpoint.py

* ``ref_ckpt_best.pt``  -- ``torch.save(model.state_dict(), path)``, the ModelCheckpoint callback's format
  (vitm/training/callbacks.py:28-33);
* ``ref_ckpt_final.pt`` -- ``{'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict'}``, the end-of-
  training format (vitm/training/train.py:245-249), with the optimizer and scheduler of
  ``getOptimizerAndSchedular`` (train.py:65-93: AdamW lr 1e-3, wd 0.05, betas (0.9, 0.98); LinearLR 0.01 -> 1 over
  500 steps, then CosineAnnealingLR to 1e-6) after two optimisation steps.
``vitm.training`` itself is not importable here (it needs torchvision), so the optimizer / scheduler are built
from those lines' arguments; the model is the reference's VITC(VisionTransformer) on a very small configuration.
"""
from __future__ import annotations

import contextlib
import functools
import io
import os
import sys

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from vitm.models import PatchEmbed, VisionTransformer, VITC  # noqa: E402

CKPT_CFG = dict(image_size=16, patch_size=8, num_classes=5, num_heads=2, mlp_ratio=2.0, depth=2, embed_dim=64)


def main():
    torch.manual_seed(1234)
    LN = nn.LayerNorm
    c = CKPT_CFG
    vt = VisionTransformer(c["image_size"], patch_size=c["patch_size"], num_classes=c["num_classes"],
                           num_heads=c["num_heads"], mlp_ratio=c["mlp_ratio"], depth=c["depth"],
                           embed_dim=c["embed_dim"], norm_layer=LN, embed_norm_layer=LN, final_norm_layer=LN,
                           embed_layer=functools.partial(PatchEmbed, embed_dim=c["embed_dim"]))
    model = VITC(vt)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    sched = torch.optim.lr_scheduler.SequentialLR(
        opt, [torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, end_factor=1.0, total_iters=500),
              torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=1000 - 500, eta_min=1e-6)], milestones=[500])
    x = torch.randn(4, 3, 16, 16)
    y = torch.tensor([0, 1, 2, 3])
    for _ in range(2):   # two steps so that exp_avg / exp_avg_sq / step exist
        with contextlib.redirect_stdout(io.StringIO()):
            f = vt.forward_features(x)
        for blk in vt.blocks:
            f = blk(f)
        logits = model.head(vt.norm_final(f)[:, 0])
        loss = nn.functional.cross_entropy(logits, y)
        opt.zero_grad()
        loss.backward()
        nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        sched.step()
    torch.save(model.state_dict(), os.path.join(HERE, "ref_ckpt_best.pt"))
    torch.save({"model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict(),
                "scheduler_state_dict": sched.state_dict()}, os.path.join(HERE, "ref_ckpt_final.pt"))
    for n in ("ref_ckpt_best.pt", "ref_ckpt_final.pt"):
        print(n, os.path.getsize(os.path.join(HERE, n)) // 1024, "KiB")


if __name__ == "__main__":
    main()
