"""Large-sample logit fixtures for the bf16 tolerance test (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_logit_stats.py

The 10-class fixtures of make_golden.py hold 10-40 logits each, too few for a max-abs error statistic to be
stable (the maximum of 20 bf16-rounding errors moves by +-30 % from one sample to the next).  This script runs
the IMPORTED reference (same recipe as make_golden.py) on a batch of 64 images per small configuration and stores
only the fp32 logits and the reference's own bf16 logits (module.bfloat16(), bf16 input): 640 logits each.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import weights as W  # noqa: E402
from make_golden import build_reference, run_reference  # noqa: E402  (imports vitm from /root/reference)


def gen(name: str, batch: int, seed: int):
    cfg = W.CONFIGS[name]
    sd = W.make_state_dict(cfg, seed)
    model = build_reference(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    x = torch.from_numpy(W.make_input(cfg, batch, seed + 1))
    _, _, logits = run_reference(model, x)
    _, _, logits_bf16 = run_reference(model.bfloat16(), x.bfloat16())
    d = (logits_bf16.float() - logits)
    path = os.path.join(HERE, f"logit_stats_{name}.npz")
    np.savez_compressed(path, logits=logits.numpy(), logits_ref_bf16=logits_bf16.float().numpy(),
                        batch=np.int64(batch), seed=np.int64(seed))
    print(f"{name}: {path} ({os.path.getsize(path) / 1024:.0f} KiB) max|logit| {logits.abs().max():.3f}; reference "
          f"bf16-vs-fp32: max {d.abs().max():.4f} rms {d.pow(2).mean().sqrt():.4f}")


if __name__ == "__main__":
    gen("tiny2", batch=64, seed=20)
    gen("vit_tiny_cifar", batch=64, seed=21)
    gen("cifar_recipe", batch=64, seed=22)
