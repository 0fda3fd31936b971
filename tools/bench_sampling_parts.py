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
    # the same with every part's stream bound to its own set of CUs (hipExtStreamCreateWithCUMask): does the longest chain lose its time
    # to the short chains' workgroups sitting on the CUs it wants, or to the dispatch path?
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    def masked_stream(lo, hi):
        words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if lo <= 32 * w + b < hi) for w in range(8)])
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(st.value)
    for split in ((0, 160, 208, 256), (0, 192, 224, 256), (0, 128, 192, 256)):
        dp._part_streams = [masked_stream(split[i], split[i + 1]) for i in range(3)]
        for p_ in dp.priors:                              # the step graphs were captured on the old streams: re-capture
            for k in [k for k in vars(p_.diffusion) if "graph" in k.lower() and k != "use_graph"]:
                try:
                    setattr(p_.diffusion, k, None if not isinstance(getattr(p_.diffusion, k), dict) else {})
                except Exception:
                    pass
        try:
            print(f"three streams on CU sets {split}: {timed(lambda: dp.sample_latents(B)):.3f} s", flush=True)
        except Exception as e:
            print(f"CU sets {split}: failed ({type(e).__name__}: {e})", flush=True)
