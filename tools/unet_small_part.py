#!/usr/bin/env python3
"""One U-Net call of a small MNIST latent part as graph replays (for tools/kprof.sh): argv = batch, side, channels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
from diffusion_prior import Unet
b, hw, ch = (int(v) for v in (sys.argv[1:4] or (64, 4, 16)))
dev = torch.device("cuda", 0)
with torch.no_grad():
    u = Unet(dim=64, dim_mults=(1, 2), channels=ch, resnet_block_groups=8).to(dev)
    u.enable_time_table(1000)
    x = torch.randn(b, hw, hw, ch, device=dev)
    t = torch.randint(0, 1000, (b,), device=dev)
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
    for _ in range(400): g.replay()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 400 * 1e6:.1f} us per U-Net call ({b}, {hw}x{hw}x{ch})")
