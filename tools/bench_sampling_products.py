import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import nfdp_hip as K
from diffusion_prior import DiffusionPrior, IdentityFormater
B, T = 64, 1000
fmt = IdentityFormater(3, 1, 32)
shapes = fmt.get_input_shapes()
with torch.no_grad():
    for mode in ("exact", "bf16x6"):
        K.set_f32_products(mode)
        torch.manual_seed(0)
        prior = DiffusionPrior(latent_formater=fmt, unet_kwargs=dict(dim=64, dim_mults=(1, 2), resnet_block_groups=8),
                               diffusion_kwargs=dict(timesteps=T, sampling_timesteps=T, loss_type="l2", beta_schedule="cosine"),
                               image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes])
        prior.concurrent_parts = True
        prior.sample_latents(B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        zs = prior.sample_latents(B)
        torch.cuda.synchronize()
        print(mode, f"{time.perf_counter() - t0:.3f} s", [float(z.abs().mean()) for z in zs], flush=True)
K.set_f32_products("exact")
