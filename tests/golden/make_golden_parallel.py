"""Golden fixtures for the ViT-22B style ``ParallelBlock`` from the IMPORTED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_parallel.py

The reference module (vitm/models/parallel_block.py:129-221) is run stand-alone in fp32 on the CPU (its forward
prints, parallel_block.py:219); weights come from a frozen numpy stream and are stored in the fixture.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from vitm.models.parallel_block import ParallelBlock  # noqa: E402

CASES = {  # name: (dim, heads, mlp_ratio, mlp_bias, tokens, batch, seed)
    "pblock_d128": (128, 2, 0.7, False, 65, 2, 31),      # hidden 89: N and K padding
    "pblock_d192": (192, 3, 1.0, True, 50, 2, 32),
}


def gen(name):
    dim, heads, ratio, mlp_bias, ntok, batch, seed = CASES[name]
    rs = np.random.RandomState(seed)
    m = ParallelBlock(dim, heads, mlp_ratio=ratio, mlp_bias=mlp_bias).eval()
    sd = {}
    for k, v in m.state_dict().items():
        if k.endswith("norm.weight"):
            a = 1.0 + 0.2 * rs.standard_normal(v.shape)
        elif k == "layer_scale.gamma":
            a = 1.0 + 0.1 * rs.standard_normal(v.shape)
        elif k.endswith("bias"):
            a = 0.2 * rs.standard_normal(v.shape)
        elif k == "fused_proj_1.weight":
            a = 1.5 * rs.standard_normal(v.shape) / np.sqrt(v.shape[1])
        else:
            a = rs.standard_normal(v.shape) / np.sqrt(v.shape[1])
        sd[k] = a.astype(np.float32)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    x = rs.standard_normal((batch, ntok, dim)).astype(np.float32)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        y = m(torch.from_numpy(x)).numpy()
        y_bf16 = m.bfloat16()(torch.from_numpy(x).bfloat16()).float().numpy()
    out = {"x": x, "y": y, "y_ref_bf16": y_bf16, "heads": np.int64(heads), "mlp_ratio": np.float64(ratio),
           "mlp_bias": np.int64(mlp_bias)}
    out.update({"sd." + k: v for k, v in sd.items()})
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), keys {list(sd)}, max|y|={np.abs(y).max():.3f}, "
          f"reference bf16-vs-fp32 err {np.abs(y_bf16 - y).max():.4f}")


if __name__ == "__main__":
    for n in CASES:
        gen(n)
