#!/usr/bin/env python3
"""A/B of the U-Net fusion switches on the largest CelebA latent part (B = 32, 6 x 64 x 64, dim 64, mults [1,2,2], groups 16)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import nfdp_hip as K
from diffusion_prior import Unet
dev = torch.device("cuda", 0)
shape = tuple(int(v) for v in (sys.argv[1:4] or (32, 64, 6)))
with torch.no_grad():
    u = Unet(dim=64, dim_mults=(1, 2, 2), channels=shape[2], resnet_block_groups=16).to(dev)
    x = torch.randn(shape[0], shape[1], shape[1], shape[2], device=dev)
    t = torch.randint(0, 1000, (shape[0],), device=dev)
    for name, setter in (("all on", None), ("GroupNorm carried off", K.set_conv_gn_carried), ("res_conv fused off", K.set_conv_res_fused),
                         ("qkv in attention off", K.set_attention_qkv_fused), ("all on", None)):
        if setter: setter(False)
        for _ in range(2): u.forward_nhwc(x, t)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            u.forward_nhwc(x, t)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                y = u.forward_nhwc(x, t)
        torch.cuda.synchronize()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40): g.replay()
        torch.cuda.synchronize()
        print(f"{name:24s}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per U-Net call {shape}", flush=True)
        if setter: setter(True)
