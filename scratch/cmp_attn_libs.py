"""Attention output of the loaded library against fp64 softmax attention on the same bf16 inputs; saves / compares."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops
torch.manual_seed(0)
B, N, H, hd = 8, 577, 12, 64
D = H * hd
qkv = (torch.randn(B * N, 3 * D, device="cuda") * 1.2).to(torch.bfloat16)
out = torch.empty((B * N, D), dtype=torch.bfloat16, device="cuda")
ops.attention(qkv, out, B, N, H, hd, hd ** -0.5)
torch.cuda.synchronize()
q, k, v = qkv.double().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).transpose(1, 2).reshape(B * N, D)
e = (out.double() - ref)
name = os.path.basename(os.environ.get("VITM_GFX950_LIB", "default"))
print(name, "rms err %.6g  max err %.6g  (ref rms %.4g)" % (e.pow(2).mean().sqrt().item(), e.abs().max().item(), ref.pow(2).mean().sqrt().item()))
f = "/tmp/attn_out_cmp.pt"
if os.path.exists(f):
    other = torch.load(f)
    d = (out.float().cpu() - other.float())
    print("   vs previous library: differing elements", int((d != 0).sum()), "of", d.numel(), " max", d.abs().max().item())
torch.save(out.cpu(), f)
