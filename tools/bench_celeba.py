#!/usr/bin/env python3
"""BASELINE configs[4] shape on one GPU: Glow L=5 K=16 on 128x128x3, B=32 (forward+NLL, inverse, sampling inverse) and one
reverse-diffusion step of the per-part U-Nets (dim 64, mults [1,2,2], groups 16).  Geometries outside the fused flow-step
kernels' table (they cover the 32x32 models) run on the per-op kernels.  usage: python tools/bench_celeba.py [B] [compute]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import Glow, GaussianPrior, calculate_output_shapes
from diffusion_prior import Unet, GaussianDiffusion

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
compute = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda", 0)


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    flow = Glow(in_channel=3, L=5, K=16, compute=compute)
    prior = GaussianPrior(2 ** 6 * 3)
    flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 1, DEEP))) for k, v in flow.state_dict().items()})
    prior.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 2, DEEP))) for k, v in prior.state_dict().items()})
    flow.to(dev); prior.to(dev)
    g = torch.Generator().manual_seed(0)
    x = ((torch.randint(0, 256, (B, 3, 128, 128), generator=g) // 8).float() / 32.0 - 0.5 + torch.rand(B, 3, 128, 128, generator=g) / 32.0).to(dev)
    xh = K.to_nhwc(x).contiguous()

    def fwd():
        ldj = torch.zeros(B, dtype=torch.float64, device=dev); logp = torch.zeros(B, dtype=torch.float64, device=dev)
        zs = flow.transform_nhwc(xh, ldj, logp)
        prior.accumulate_log_prob_nhwc(zs[-1], logp)
        return zs, ldj, logp

    zs, ldj, logp = fwd()
    bpd = float((-(ldj + logp) / (np.log(2.0) * 128 * 128 * 3) + 5.0).mean())
    from normalizing_flow import glow as _glow
    _glow.set_weight_term_cache(False)
    t = timeit(fwd)
    print(f"Glow L5 K16 128x128x3 B={B} {compute}: forward+NLL {t * 1e3:.2f} ms = {B / t:.0f} img/s (bits/dim {bpd:.4f}), "
          f"log-dets recomputed every call as in the reference", flush=True)
    t = timeit(lambda: flow.invert_nhwc([z.contiguous() for z in zs]))
    print(f"  inverse {t * 1e3:.2f} ms = {B / t:.0f} img/s, W^-1 recomputed every call", flush=True)
    _glow.set_weight_term_cache(True)
    t = timeit(fwd)
    print(f"  with the weight-only terms kept until a weight changes (default): forward+NLL {t * 1e3:.2f} ms = {B / t:.0f} img/s", flush=True)
    _glow.set_weight_term_cache(False)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fwd()
    t = timeit(gr.replay)
    print(f"  forward+NLL replayed as one hipGraph (log-dets recomputed inside): {t * 1e3:.2f} ms = {B / t:.0f} img/s", flush=True)
    _glow.set_weight_term_cache(True)
    lat = [z.contiguous() for z in zs]
    t = timeit(lambda: flow.invert_nhwc(lat))
    rec = flow.invert_nhwc(lat)
    print(f"  inverse {t * 1e3:.2f} ms = {B / t:.0f} img/s, round trip max rel err {float((rec - xh).abs().max() / xh.abs().max()):.2e}", flush=True)
    t = timeit(lambda: flow.invert_nhwc([lat[-1]], temperature=0.7))
    print(f"  sampling inverse (T=0.7) {t * 1e3:.2f} ms = {B / t:.0f} img/s", flush=True)
    for (c, h, w) in calculate_output_shapes(5, 3, 128):
        u = Unet(dim=64, dim_mults=(1, 2, 2), channels=int(c), resnet_block_groups=16)
        d = GaussianDiffusion(u, image_size=int(h), timesteps=50, beta_schedule="cosine", loss_type="l2", auto_normalize=False).to(dev)
        d.sample(batch_size=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.sample(batch_size=B)
        torch.cuda.synchronize()
        print(f"  U-Net reverse step, part {int(c)}x{int(h)}x{int(w)} B={B}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms", flush=True)
