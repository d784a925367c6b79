import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch
from vitm_amd import _ops as ops, _lib
lib = _lib.load()
M, N, K = 50432, int(os.environ.get("N", 3072)), int(os.environ.get("K", 768))
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = torch.randn(N, K, device="cuda") * K ** -0.5
pk = ops.pack_linear(W, torch.zeros(N, device="cuda"))
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
tiles = ((M + 255)//256) * ((N+255)//256)
dbg = torch.zeros((tiles, 8), dtype=torch.int64, device="cuda")
def run():
    _lib.check(lib.vitm_gemm(ops._p(A), K, ops._p(pk.wp), pk.kpad, ops._p(out), N, 0, M, N, K, 0, None, None, ops._p(pk.c), None, ops._p(dbg), N, 0, ops._stream()), "gemm")
for _ in range(3): run()
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype("float64")
t0 = d[:, 0].min()
d = (d - t0)  # ticks (100 MHz?) s_memtime is shader clock cycles
names = ["start", "prologue_done", "mainloop_done", "epi_lds_done", "stores_issued", "stores_complete"]
import numpy as np
order = np.argsort(d[:, 0])
print("tiles", tiles, "kernel span (cycles)", d[:, 5].max())
for k in range(1, 6):
    seg = d[:, k] - d[:, k-1]
    print(f"{names[k]:16s} mean {seg.mean():9.0f} med {np.median(seg):9.0f} min {seg.min():9.0f} max {seg.max():9.0f}")
# rounds: start times histogram
st = np.sort(d[:, 0])
print("start times (every 256th):", st[::256].astype(int))
print("per-tile total mean", (d[:,5]-d[:,0]).mean())
