"""Experiment: batch split across S HIP streams so one chunk's kernel tails overlap the other chunk's kernels."""
import sys, time, torch, torch.nn as nn
sys.path.insert(0, "vision-transformer_amd")
from vitm_amd.models import VisionTransformer, VITC
vt = VisionTransformer(224, patch_size=16, num_classes=1000, num_heads=12, mlp_ratio=4.0, depth=12, embed_dim=768,
                       norm_layer=nn.LayerNorm, final_norm_layer=nn.LayerNorm)
m = VITC(vt).eval().cuda().to(torch.bfloat16)
B = 256
x = torch.randn(B, 3, 224, 224, device="cuda", dtype=torch.bfloat16)
def run(S, steps):
    streams = [torch.cuda.Stream() for _ in range(S)]
    chunks = x.chunk(S)
    cur = torch.cuda.current_stream()
    for _ in range(steps):
        for s, c in zip(streams, chunks):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                m(c)
        for s in streams:
            cur.wait_stream(s)
with torch.no_grad():
    for S in (1, 2, 3, 4):
        run(S, 3); torch.cuda.synchronize()
        t = time.time(); run(S, 20); torch.cuda.synchronize(); dt = (time.time() - t) / 20
        print(f"streams={S}: {dt*1e3:.2f} ms/step {B/dt:.0f} img/s", flush=True)
