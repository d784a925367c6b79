"""Golden fixture of the training step from the IMPORTED reference modules (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py

The loop body of vitm/training/train.py:215-224 -- zero_grad, forward, nn.CrossEntropyLoss, backward,
clip_grad_norm_(1.0), AdamW step (lr 1e-3, wd 0.05, betas (0.9, 0.98), train.py:68-72), scheduler step (LinearLR
0.01 -> 1 over 500 iterations then cosine, train.py:75-91) -- run for three steps in fp32 on the CPU with the reference's
VITC(VisionTransformer) on the ``tiny2`` configuration (all optional norms on) and the deterministic synthetic
weights of tests/golden/weights.py.  ``vitm.training`` itself needs torchvision and is not importable; the loop is
restated from those lines.  Stored: per-step loss and pre-clip gradient norm, every parameter's gradient of step 1
(raw, before clipping), every parameter after step 3.
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

NAME, SEED, BATCH, STEPS = "tiny2", 41, 6, 3


def forward(model, x):
    vt = model.vit_model
    with contextlib.redirect_stdout(io.StringIO()):
        f = vt.forward_features(x)
    for blk in vt.blocks:          # VisionTransformer.forward calls the ModuleList (models.py:409): iterate instead
        f = blk(f)
    return model.head(vt.norm_final(f)[:, 0])


def main():
    cfg = W.CONFIGS[NAME]
    LN = nn.LayerNorm
    vt = VisionTransformer(cfg["image_size"], patch_size=cfg["patch_size"], num_classes=cfg["num_classes"],
                           num_heads=cfg["num_heads"], mlp_ratio=cfg["mlp_ratio"], depth=cfg["depth"],
                           embed_dim=cfg["embed_dim"], norm_layer=LN, embed_norm_layer=LN, final_norm_layer=LN,
                           embed_layer=functools.partial(PatchEmbed, embed_dim=cfg["embed_dim"], bias=cfg["patch_bias"]))
    model = VITC(vt).train()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, SEED).items()}, strict=True)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    sched = torch.optim.lr_scheduler.SequentialLR(
        opt, [torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, end_factor=1.0, total_iters=500),
              torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=1000 - 500, eta_min=1e-6)], milestones=[500])
    crit = nn.CrossEntropyLoss()
    rs = np.random.RandomState(SEED + 1)
    out = {"seed": np.int64(SEED), "batch": np.int64(BATCH), "steps": np.int64(STEPS)}
    losses, norms, lrs = [], [], []
    for step in range(STEPS):
        x = torch.from_numpy(W.make_input(cfg, BATCH, SEED + 10 + step))
        y = torch.from_numpy(rs.randint(0, cfg["num_classes"], size=BATCH).astype(np.int64))
        out[f"labels{step}"] = y.numpy()
        opt.zero_grad()
        loss = crit(forward(model, x), y)
        loss.backward()
        if step == 0:
            for k, p in model.named_parameters():
                out["grad." + k] = p.grad.detach().numpy().copy()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        losses.append(loss.item())
        norms.append(norm.item())
    out["losses"], out["grad_norms"], out["lrs"] = np.array(losses), np.array(norms), np.array(lrs)
    for k, p in model.named_parameters():
        out["param." + k] = p.detach().numpy().copy()
    path = os.path.join(HERE, "train_tiny2.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB): losses {losses}, grad norms {norms}, lrs {lrs}")


if __name__ == "__main__":
    main()
