#!/usr/bin/env python3
"""Host cost of replaying the captured reverse-step graphs (BASELINE configs[3] parts): time to ENQUEUE 300 replays (no sync) against
the time the GPU takes to run them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
from diffusion_prior import DiffusionPrior, IdentityFormater
Bs, T = 64, 1000
fmt = IdentityFormater(3, 1, 32)
shapes = fmt.get_input_shapes()
dp = DiffusionPrior(latent_formater=fmt, unet_kwargs=dict(dim=64, dim_mults=(1, 2), resnet_block_groups=8),
                    diffusion_kwargs=dict(timesteps=T, sampling_timesteps=T, loss_type="l2", beta_schedule="cosine"),
                    image_sizes=[int(sh[1]) for sh in shapes], latent_channels=[int(sh[0]) for sh in shapes])
with torch.no_grad():
    dp.sample_latents(Bs)
    torch.cuda.synchronize()
    for pr, sh in zip(dp.priors, shapes):
        d = pr.diffusion if hasattr(pr, "diffusion") else pr
        keys = [k for k in d._graphs if k[0] == "step"]
        g, xh, tb, gen = d._graphs[keys[0]]
        tb.fill_(900)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"part {tuple(int(v) for v in sh)}: enqueue {1e6 * (t1 - t0) / 300:.0f} us per replay on the host, {1e6 * (t2 - t0) / 300:.0f} us per replay until the GPU is done", flush=True)
