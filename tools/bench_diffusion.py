#!/usr/bin/env python3
"""Reverse-diffusion step latency (BASELINE configs[3] shape: MNIST latent parts, U-Net dim 64 mults [1,2] groups 8):
eager launches vs the hipGraph-captured step.  usage: python tools/bench_diffusion.py [batch] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
from diffusion_prior import Unet, GaussianDiffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
with torch.no_grad():
    for ch, size in ((2, 16), (4, 8), (16, 4)):
        u = Unet(dim=64, dim_mults=(1, 2), channels=ch, resnet_block_groups=8)
        d = GaussianDiffusion(u, image_size=size, timesteps=T, beta_schedule="cosine", loss_type="l2", auto_normalize=False).to(dev)
        res = {}
        for mode in (False, True):
            d.use_graph = mode
            d.sample(batch_size=B)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s = d.sample(batch_size=B)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / T * 1e3
            assert torch.isfinite(s).all()
        print(f"latent {ch}x{size}x{size} B={B}: eager {res[False]:.3f} ms/step, hipGraph {res[True]:.3f} ms/step "
              f"({res[False] / res[True]:.2f}x)", flush=True)
