"""ViT-B/16 fp32-path throughput (informational)."""
import sys, time, torch, torch.nn as nn
sys.path.insert(0, "vision-transformer_amd")
from vitm_amd.models import VisionTransformer, VITC
vt = VisionTransformer(224, patch_size=16, num_classes=1000, num_heads=12, mlp_ratio=4.0, depth=12, embed_dim=768,
                       norm_layer=nn.LayerNorm, final_norm_layer=nn.LayerNorm)
m = VITC(vt).eval().cuda()
B = 64
x = torch.randn(B, 3, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(2): m(x)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): m(x)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
print(f"fp32 ViT-B/16 B={B}: {dt*1e3:.1f} ms/step, {B/dt:.0f} img/s, {B*35.128e9/dt/1e12:.1f} TFLOP/s", flush=True)
