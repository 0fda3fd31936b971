#!/usr/bin/env python3
"""One reverse chain of T captured steps for a single latent part (for rocprofv3 --kernel-trace --stats).
usage: python tools/run_sampling_only.py [batch] [T] [channels] [size] [compute f32|bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
from diffusion_prior import Unet, GaussianDiffusion
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
size = int(sys.argv[4]) if len(sys.argv) > 4 else 16
compute = sys.argv[5] if len(sys.argv) > 5 else "f32"
dev = torch.device("cuda", 0)
with torch.no_grad():
    u = Unet(dim=64, dim_mults=(1, 2), channels=ch, resnet_block_groups=8, compute=compute)
    d = GaussianDiffusion(u, image_size=size, timesteps=T, beta_schedule="cosine", loss_type="l2", auto_normalize=False).to(dev)
    d.sample(batch_size=B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.sample(batch_size=B)
    torch.cuda.synchronize()
    print(f"B={B} T={T} part {ch}x{size}x{size} {compute}: {(time.perf_counter() - t0) / T * 1e3:.3f} ms per timestep")
