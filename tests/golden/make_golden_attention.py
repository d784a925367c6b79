"""Golden fixtures for the non-default Attention variants, from the IMPORTED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_attention.py

``Attention(qkv_bias=True, qk_norm=True)`` (vitm/models/models.py:32-35,54-55,82): per-head LayerNorm of q
and k over head_dim before the scaling.  Weights come from a frozen numpy stream (stored in the fixture: they
are small), the output is the reference module's fp32 CPU forward.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from vitm.models.models import Attention  # noqa: E402

CASES = {  # name: (dim, heads, tokens, batch, qkv_bias, qk_norm, seed)
    "attn_qknorm_hd64": (128, 2, 65, 2, True, True, 21),
    "attn_qknorm_hd96": (192, 2, 50, 2, False, True, 22),
    "attn_qkvbias_hd32": (64, 2, 33, 3, True, False, 23),
}


def gen(name):
    dim, heads, ntok, batch, qkv_bias, qk_norm, seed = CASES[name]
    rs = np.random.RandomState(seed)
    m = Attention(dim, num_heads=heads, qkv_bias=qkv_bias, qk_norm=qk_norm).eval()
    sd = {}
    for k, v in m.state_dict().items():
        if k.endswith("norm.weight"):
            a = 1.0 + 0.2 * rs.standard_normal(v.shape)
        elif k.endswith("bias"):
            a = 0.2 * rs.standard_normal(v.shape)
        elif k == "qkv.weight":
            a = 2.0 * rs.standard_normal(v.shape) / np.sqrt(dim)
        else:
            a = rs.standard_normal(v.shape) / np.sqrt(dim)
        sd[k] = a.astype(np.float32)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    x = rs.standard_normal((batch, ntok, dim)).astype(np.float32)
    with torch.no_grad():
        y = m(torch.from_numpy(x)).numpy()
        y_bf16 = m.bfloat16()(torch.from_numpy(x).bfloat16()).float().numpy()
    out = {"x": x, "y": y, "y_ref_bf16": y_bf16, "heads": np.int64(heads), "qkv_bias": np.int64(qkv_bias),
           "qk_norm": np.int64(qk_norm)}
    out.update({"sd." + k: v for k, v in sd.items()})
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), max|y|={np.abs(y).max():.3f}, "
          f"reference bf16-vs-fp32 err {np.abs(y_bf16 - y).max():.4f}")


if __name__ == "__main__":
    for n in CASES:
        gen(n)
