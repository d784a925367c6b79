import os, sys
ROOT = "/root/repo"
ROOT = os.environ.get("GRAFT_REPO_ROOT", ROOT)
for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd")): sys.path.insert(0, p)
import torch, numpy as np
from vitm_amd import _ops as ops, _lib
lib = _lib.load()
M = 50432
def run(name, N, K, epi):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda") * 0.1
    pk = ops.pack_linear(W, b)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    dbg = torch.zeros((256, 12, 8, 20), dtype=torch.int64, device="cuda")
    for _ in range(20):
        _lib.check(lib.vitm_gemm(ops._p(A), K, ops._p(pk.wp), pk.kpad, ops._p(out), N, 0, M, N, K, epi, None, ops._p(pk.s), ops._p(pk.c), None, ops._p(dbg), N, 0, ops._stream()), "gemm")
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype("float64")
    # slot 3 = realtime (100 MHz) at tile start, wave 0; slot 0 = cycles at tile start; slot 1 K loop end; slot 2 epilogue end
    for it in (0, 1, 3, 6):
        rt = d[:, it, 0, 3]
        cls = (np.arange(256) >> 3) % 2
        print(name, "tile", it, "realtime tile start, class 1 - class 0 (us):", (np.median(rt[cls == 1]) - np.median(rt[cls == 0])) / 100.0,
              " epilogue cycles group0/group1, class 0:", np.median(d[cls == 0, it, 0, 2] - d[cls == 0, it, 0, 1]), np.median(d[cls == 0, it, 4, 2] - d[cls == 0, it, 4, 1]),
              " class 1:", np.median(d[cls == 1, it, 0, 2] - d[cls == 1, it, 0, 1]), np.median(d[cls == 1, it, 4, 2] - d[cls == 1, it, 4, 1]))
run("plain3072", 3072, 768, ops.EPI_BIAS)
