"""What streaming kernels reach on this box (MI355X), as a yardstick for the HBM-bound kernels: device-to-device copy
(hipMemcpyAsync) and torch's read-only sum / elementwise copy kernels on buffers larger than the 256 MB Infinity Cache."""
import torch, time
def bench(f, nbytes, name, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"{name:40s} {nbytes / dt / 1e12:6.2f} TB/s  ({dt * 1e6:7.1f} us)")
for mb in (310, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n, device="cuda", dtype=torch.bfloat16); b = torch.empty_like(a)
    bench(lambda: b.copy_(a), 2 * n * 2, f"copy {mb} MB (read + write)")
    bench(lambda: a.sum(), n * 2, f"sum {mb} MB (read only)")
    bench(lambda: b.fill_(1.0), n * 2, f"fill {mb} MB (write only)")
    # 3:1 read:write like attention (read q,k,v; write o)
    c = torch.empty(n // 3, device="cuda", dtype=torch.bfloat16)
    bench(lambda: torch.add(a[:n // 3], a[n // 3:2 * (n // 3)], out=c), (n // 3) * 2 * 3, f"add {mb} MB (2 reads + 1 write)")
