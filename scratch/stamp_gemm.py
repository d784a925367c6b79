import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch, numpy as np
from vitm_amd import _ops as ops, _lib
lib = _lib.load()
M, N, K = 50432, int(os.environ.get("N", 3072)), int(os.environ.get("K", 768))
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = torch.randn(N, K, device="cuda") * K ** -0.5
pk = ops.pack_linear(W, torch.zeros(N, device="cuda"))
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
dbg = torch.zeros((256, 12, 4), dtype=torch.int64, device="cuda")
def run():
    _lib.check(lib.vitm_gemm(ops._p(A), K, ops._p(pk.wp), pk.kpad, ops._p(out), N, 0, M, N, K, 0, None, None, ops._p(pk.c), None, ops._p(dbg), N, 0, ops._stream()), "gemm")
for _ in range(3): run()
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype("float64")   # [wg, it, (start, mainloop_end, epi_end, realtime_start)]
nt = 9
wg = 5
print("WG", wg, "per tile: mainloop, epilogue, gap-to-next-start (cycles)")
for it in range(nt):
    print(it, int(d[wg, it, 1] - d[wg, it, 0]), int(d[wg, it, 2] - d[wg, it, 1]), int(d[wg, it+1, 0] - d[wg, it, 2]) if it + 1 < nt else -1)
ml = d[:, :nt, 1] - d[:, :nt, 0]; ep = d[:, :nt, 2] - d[:, :nt, 1]
print("mean mainloop per it:", ml.mean(0).astype(int))
print("mean epilogue per it:", ep.mean(0).astype(int))
cyc = d[:, nt-1, 0] - d[:, 0, 0]; rt = (d[:, nt-1, 3] - d[:, 0, 3])   # realtime ticks @100MHz
print("clock GHz (median):", np.median(cyc / (rt * 10.0)))
print("per-tile total cycles (mean):", (cyc / (nt - 1)).mean())
