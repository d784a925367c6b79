"""Per-tile timeline of the 256x256 GEMM from a -DPP_STAMP build (VITM_GFX950_LIB=.../libvitm_gfx950_stamp.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch, numpy as np
from vitm_amd import _ops as ops, _lib
lib = _lib.load()
M = 50432
PA = int(os.environ.get("PAD_A", "0")); PW = int(os.environ.get("PAD_W", "0"))   # extra elements of row pitch
def stamp(name, N, K, epi):
    A = torch.randn(M, K + PA, device="cuda").to(torch.bfloat16)[:, :K]
    W = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda") * 0.1
    if epi in (ops.EPI_LN, ops.EPI_LN_GELU):
        pk = ops.pack_linear(W, b, torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"))
        stats = ops.ln_stats(A, M, K, 1e-5)
    else:
        pk = ops.pack_linear(W, b); stats = None
    wp = pk.wp
    if PW:
        wp = torch.zeros((pk.wp.shape[0], pk.kpad + PW), dtype=pk.wp.dtype, device="cuda"); wp[:, :pk.kpad] = pk.wp
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    dbg = torch.zeros((256, 12, 20), dtype=torch.int64, device="cuda")
    def run():
        _lib.check(lib.vitm_gemm(ops._p(A), K + PA, ops._p(wp), pk.kpad + PW, ops._p(out), N, 0, M, N, K, epi, ops._p(stats), ops._p(pk.s), ops._p(pk.c), None, ops._p(dbg), N, 0, ops._stream()), "gemm")
    for _ in range(20): run()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype("float64")
    ONLY = os.environ.get("ONLY")
    tiles = (M + 255) // 256 * (N // 256)
    G = int(os.environ.get("VITM_PP_GRID", "256"))
    nt = min(tiles // G, 11)
    d = d[:G]
    ml = d[:, :nt, 1] - d[:, :nt, 0]; ep = d[:, :nt, 2] - d[:, :nt, 1]
    gap = d[:, 1:nt, 0] - d[:, :nt-1, 2]
    k0 = d[:, :nt, 4] - d[:, :nt, 0]; k1 = d[:, :nt, 5] - d[:, :nt, 4]; k2 = d[:, :nt, 6] - d[:, :nt, 5]
    rest = (d[:, :nt, 1] - d[:, :nt, 6]) / (K // 64 - 3)
    nkt = min(K // 64, 16)
    per = np.stack([d[:, 1:nt - 1, 4 + t] - (d[:, 1:nt - 1, 4 + t - 1] if t else d[:, 1:nt - 1, 0]) for t in range(nkt)], -1)
    if nt > 2: print("every K-tile of the middle tiles (median cycles):", np.median(per.reshape(-1, nkt), 0).astype(int))
    print(f"== pad A {PA} W {PW} {name} N={N} K={K} epi={epi} tiles={tiles} whole rounds stamped={nt}")
    print("median K-loop cycles per tile index:", np.median(ml, 0).astype(int))
    print("median epilogue cycles per tile index:", np.median(ep, 0).astype(int))
    print("K-tile 0 / 1 / 2 / mean of the rest (median cycles):", np.median(k0, 0).astype(int), np.median(k1, 0).astype(int), np.median(k2, 0).astype(int), np.median(rest, 0).astype(int))
    cyc = d[:, nt-1, 0] - d[:, 0, 0]; rt = d[:, nt-1, 3] - d[:, 0, 3]
    print("clock GHz (median):", np.median(cyc / (rt * 10.0)), " per-tile total cycles (median):", np.median(cyc / (nt - 1)))
ONLY = os.environ.get("ONLY")
for name, N, K, epi in (("fc1", 3072, 768, ops.EPI_LN_GELU), ("qkv", 2304, 768, ops.EPI_LN), ("plain3072", 3072, 768, ops.EPI_BIAS),
                        ("plain768", 768, 768, ops.EPI_BIAS), ("plainK3072", 768, 3072, ops.EPI_BIAS)):
    if ONLY is None or name in ONLY.split(","): stamp(name, N, K, epi)
