#!/usr/bin/env python3
"""Where the wall time of DiffusionPrior.sample_latents (3 MNIST latent parts, B = 64, T steps) comes from: every part's chain ALONE,
the three chains on three streams, and the same with the longest chain on a high-priority stream.  usage: [B=64] [T=1000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
from diffusion_prior import DiffusionPrior, IdentityFormater
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fmt = IdentityFormater(3, 1, 32)
shapes = fmt.get_input_shapes()


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0


with torch.no_grad():
    dp = DiffusionPrior(latent_formater=fmt, unet_kwargs=dict(dim=64, dim_mults=(1, 2), resnet_block_groups=8),
                        diffusion_kwargs=dict(timesteps=T, sampling_timesteps=T, loss_type="l2", beta_schedule="cosine"),
                        image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes])
    for i, prior in enumerate(dp.priors):
        print(f"part {i} ({tuple(int(v) for v in shapes[i])}) alone: {timed(lambda: prior.sample_latent(B)):.3f} s", flush=True)
    dp.concurrent_parts = True
    print(f"three streams: {timed(lambda: dp.sample_latents(B)):.3f} s", flush=True)
    dp._part_streams = None
    dp.prioritize_longest_chain = True
    print(f"three streams, longest chain on a high-priority stream: {timed(lambda: dp.sample_latents(B)):.3f} s", flush=True)
