#!/usr/bin/env python3
"""Per-kernel HIP-event trace of one eager reverse-diffusion step (diagnostic).
usage: python tools/trace_unet_step.py [batch] [compute] [channels] [size] [mults e.g. 1,2,2] [groups]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import nfdp_hip as K
from diffusion_prior import Unet, GaussianDiffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
compute = sys.argv[2] if len(sys.argv) > 2 else "f32"
ch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
size = int(sys.argv[4]) if len(sys.argv) > 4 else 16
mults = tuple(int(v) for v in sys.argv[5].split(",")) if len(sys.argv) > 5 else (1, 2)
groups = int(sys.argv[6]) if len(sys.argv) > 6 else 8
dev = torch.device("cuda", 0)
with torch.no_grad():
    u = Unet(dim=64, dim_mults=mults, channels=ch, resnet_block_groups=groups, compute=compute)
    d = GaussianDiffusion(u, image_size=size, timesteps=10, beta_schedule="cosine", auto_normalize=False).to(dev)
    d.use_graph = False
    x = torch.randn(B, size, size, ch, device=dev)
    for _ in range(2):
        d._p_sample_nhwc(x, 5, None, False)
    K.start_trace()
    d._p_sample_nhwc(x, 5, None, False)
    tr = K.stop_trace()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in tr:
    agg[e["label"].split(" hw")[0]][0] += e["ms"]; agg[e["label"].split(" hw")[0]][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"compute={compute} B={B} part {ch}x{size}x{size} mults {mults}: {len(tr)} launches, {tot:.3f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {v[0]*1e3:8.1f} us  x{v[1]:3d}  {k}")
