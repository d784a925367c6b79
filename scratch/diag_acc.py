import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.nn as nn
import weights as W
from oracle import vit_oracle as O
from test_gpu_model import build, cos_rows, rel_err

def run(name, pdtype, in_dtype):
    g = np.load(os.path.join(ROOT, "tests/golden", f"{name}.npz"))
    seed, batch = int(g["seed"]), int(g["batch"])
    model, sd, cfg = build(name, seed, dtype=pdtype)
    x = torch.from_numpy(W.make_input(cfg, batch, seed + 1)).cuda().to(in_dtype)
    with torch.no_grad():
        logits = model(x); emb = model.vit_model(x)
    ref = torch.from_numpy(g["logits"])
    err = (logits.float().cpu() - ref).abs().max().item()
    ge = emb.float().cpu()
    if "embeddings_rows" in g: ge = ge[:, torch.from_numpy(g["embeddings_rows"])]
    print(f"{name:16s} params={str(pdtype):15s} in={str(in_dtype):15s} logit_err={err:.4f} (max|logit| {ref.abs().max():.2f}) cos_min={cos_rows(ge, torch.from_numpy(g['embeddings'])):.6f} emb_rel={rel_err(ge, torch.from_numpy(g['embeddings'])):.4f}", flush=True)

for name in ["tiny2", "vit_tiny_cifar", "vit_s16", "vit_b16", "vit_b16_384", "cifar_recipe"]:
    for pd in (torch.bfloat16, torch.float32):
        for idt in (torch.bfloat16, torch.float32):
            try:
                run(name, pd, idt)
            except Exception as e:
                print(name, pd, idt, "FAILED", repr(e)[:300], flush=True)

# stage-wise for tiny2 with fp32 params
model, sd, cfg = build("tiny2", 0, dtype=torch.float32)
x = torch.from_numpy(W.make_input(cfg, 2, 1))
stages = []
ref_emb = O.vit_forward(x, sd, W.oracle_cfg(cfg), "vit_model.", torch.float32, stages)
vt = model.vit_model
with torch.no_grad():
    xb = x.cuda().to(torch.bfloat16)
    f = vt.forward_features(xb)
    print("features rel", rel_err(f.float().cpu(), stages[0]))
    h = f
    for i, blk in enumerate(vt.blocks):
        h = blk(h)
        print("block", i, "rel", rel_err(h.float().cpu(), stages[i+1]), "max", stages[i+1].abs().max().item())
    emb = vt.forward_heads(f)
    print("emb rel", rel_err(emb.float().cpu(), ref_emb), cos_rows(emb.float().cpu(), ref_emb))
