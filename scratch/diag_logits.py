"""HIP bf16 logits against the fp32 reference, next to the reference's own bf16 forward, on every logit fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import weights as W
from test_gpu_model import build
for fx, name in [("tiny2", "tiny2"), ("vit_tiny_cifar", "vit_tiny_cifar"), ("cifar_recipe", "cifar_recipe"), ("vit_s16", "vit_s16"),
                 ("vit_b16", "vit_b16"), ("vit_b16_384", "vit_b16_384"), ("logit_stats_tiny2", "tiny2"),
                 ("logit_stats_vit_tiny_cifar", "vit_tiny_cifar"), ("logit_stats_cifar_recipe", "cifar_recipe")]:
    g = np.load(os.path.join(ROOT, "tests/golden", fx + ".npz"))
    seed, batch = int(g["seed"]), int(g["batch"])
    model, sd, cfg = build(name, seed)
    x = torch.from_numpy(W.make_input(cfg, batch, seed + 1)).cuda().to(torch.bfloat16)
    with torch.no_grad():
        lg = model(x).float().cpu()
    ref = torch.from_numpy(g["logits"]); yb = torch.from_numpy(g["logits_ref_bf16"])
    e, y = (lg - ref), (yb - ref)
    print(f"{fx:28s} n={ref.numel():5d} HIP max {e.abs().max():.4f} rms {e.pow(2).mean().sqrt():.4f} | ref-bf16 max {y.abs().max():.4f} "
          f"rms {y.pow(2).mean().sqrt():.4f} | ratio max {e.abs().max()/y.abs().max():.3f} rms {e.pow(2).mean().sqrt()/y.pow(2).mean().sqrt():.3f}", flush=True)
