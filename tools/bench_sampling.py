#!/usr/bin/env python3
"""DiffusionPrior.sample_latents wall time (BASELINE configs[3] shape: MNIST latent parts 2x16x16 / 4x8x8 / 16x4x4, U-Net dim 64
mults [1,2] groups 8, cosine schedule): the three per-part reverse chains one after the other vs one HIP stream per part.
usage: python tools/bench_sampling.py [batch] [T] [S]   (S < T: DDIM with S sampling steps, the reference's yaml default is S = 100)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
from diffusion_prior import DiffusionPrior, IdentityFormater
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
S = int(sys.argv[3]) if len(sys.argv) > 3 else T
fmt = IdentityFormater(3, 1, 32)
shapes = fmt.get_input_shapes()
with torch.no_grad():
    prior = DiffusionPrior(latent_formater=fmt, unet_kwargs=dict(dim=64, dim_mults=(1, 2), resnet_block_groups=8),
                           diffusion_kwargs=dict(timesteps=T, sampling_timesteps=S, loss_type="l2", beta_schedule="cosine"),
                           image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes])
    for mode in (False, True):
        prior.concurrent_parts = mode
        prior.sample_latents(B)                      # captures the per-part step graphs / creates the streams
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        zs = prior.sample_latents(B)
        t_host = time.perf_counter() - t0               # the host has enqueued everything (no sync inside the loop)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert all(torch.isfinite(z).all() for z in zs)
        print(f"B={B} T={T} S={S} {'one stream per part' if mode else 'sequential parts    '}: {dt:.3f} s per sample_latents "
              f"= {dt / S * 1e3:.3f} ms per sampling step (3 parts), {B / dt:.1f} latent sets/s; host enqueue done after {t_host:.3f} s", flush=True)
