"""Per-wave timeline of a tile of the 256x256 GEMM from a -DPP_STAMP=2 build (VITM_GFX950_LIB=...): when each of the 8
waves ends its K loop, ends its epilogue and ends K-tile 0 of the next tile, relative to wave 0's tile start."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch, numpy as np
from vitm_amd import _ops as ops, _lib
lib = _lib.load()
M = 50432
def stamp(name, N, K, epi):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda") * 0.1
    if epi in (ops.EPI_LN, ops.EPI_LN_GELU):
        pk = ops.pack_linear(W, b, torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"))
        stats = ops.ln_stats(A, M, K, 1e-5)
    else:
        pk = ops.pack_linear(W, b); stats = None
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    dbg = torch.zeros((256, 12, 8, 20), dtype=torch.int64, device="cuda")
    for _ in range(20):
        _lib.check(lib.vitm_gemm(ops._p(A), K, ops._p(pk.wp), pk.kpad, ops._p(out), N, 0, M, N, K, epi, ops._p(stats), ops._p(pk.s), ops._p(pk.c), None, ops._p(dbg), N, 0, ops._stream()), "gemm")
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype("float64")
    it = 3                                     # a middle tile
    t0 = d[:, it, 0:1, 0]                      # wave 0's tile start, per workgroup
    nk = K // 64
    rows = [("tile start", d[:, it, :, 0]), ("after K-tile 0", d[:, it, :, 4]), ("after K-tile 1", d[:, it, :, 5]),
            (f"after K-tile {nk - 2}", d[:, it, :, 4 + nk - 2]), ("K loop end", d[:, it, :, 1]), ("epilogue end", d[:, it, :, 2]),
            ("next tile: after K-tile 0", d[:, it + 1, :, 4]), ("next tile: after K-tile 1", d[:, it + 1, :, 5])]
    print(f"== {name} N={N} K={K} epi={epi}: median over workgroups, cycles since wave 0 started the tile; columns = waves 0-7")
    for label, v in rows:
        print(f"{label:28s}", np.median(v - t0, 0).astype(int))
for name, N, K, epi in (("fc1", 3072, 768, ops.EPI_LN_GELU), ("qkv", 2304, 768, ops.EPI_LN), ("plain3072", 3072, 768, ops.EPI_BIAS)):
    stamp(name, N, K, epi)
