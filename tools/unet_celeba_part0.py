#!/usr/bin/env python3
"""The largest CelebA latent part's U-Net call (BASELINE configs[4]: 6 x 64 x 64 latents, dim 64, mults [1,2,2], groups 16, B = 32) as graph
replays, for rocprofv3 (tools/kprof.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
from diffusion_prior import Unet
dev = torch.device("cuda", 0)
with torch.no_grad():
    u = Unet(dim=64, dim_mults=(1, 2, 2), channels=6, resnet_block_groups=16).to(dev)
    x = torch.randn(32, 64, 64, 6, device=dev)
    t = torch.randint(0, 1000, (32,), device=dev)
    for _ in range(3): u.forward_nhwc(x, t)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        u.forward_nhwc(x, t)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            y = u.forward_nhwc(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per U-Net call (B=32, 6x64x64)")
