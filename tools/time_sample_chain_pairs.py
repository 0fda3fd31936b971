#!/usr/bin/env python3
"""Which chains of DiffusionPrior.sample_latents (BASELINE configs[3]) overlap: every subset of the three parts' reverse chains on their
streams (the host enqueues the live chains' steps round-robin, as DiffusionPrior._sample_parts does)."""
import os, sys, time, itertools
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
    graphs = []
    for pr in dp.priors:
        d = pr.diffusion if hasattr(pr, "diffusion") else pr
        k = [k for k in d._graphs if k[0] == "step"][0]
        graphs.append(d._graphs[k])
    prio = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else [0, 0, 0]      # stream priorities (-1 = high) of the three chains
    print("stream priorities", prio)
    streams = [torch.cuda.Stream(priority=p) for p in prio]
    names = ["16x16", "8x8", "4x4"]
    for r in (1, 2, 3):
        for sub in itertools.combinations(range(3), r):
            for i in sub:
                graphs[i][2].fill_(900)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                for i in sub:
                    with torch.cuda.stream(streams[i]):
                        graphs[i][0].replay()
            torch.cuda.synchronize()
            print(f"{' + '.join(names[i] for i in sub):22s}: {(time.perf_counter() - t0) / 300 * 1e6:7.1f} us per step", flush=True)
    # two copies of the same chain's graph cannot run concurrently (one set of buffers); the 16x16 chain beside an idle-spinning stream: n/a
