"""Host-side cost of an eager forward at small batch (launch-bound): cProfile over 300 forwards of ViT-tiny CIFAR, batch 64."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch, weights as W
from test_gpu_model import build
model, sd, cfg = build("vit_tiny_cifar", 5)
x = torch.from_numpy(W.make_input(cfg, 64, 9)).cuda().to(torch.bfloat16)
with torch.no_grad():
    for _ in range(20): model(x)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): model(x)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
