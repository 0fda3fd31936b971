#!/usr/bin/env python3
"""Forward+NLL step including the host->device copy of the batch (the bench's `value` starts with the batch in HBM).
uint8 images [512,3,32,32] from pinned host memory; dequantisation noise either copied too (as bench.py's fixed batch) or drawn
on the device.  usage: python tools/bench_h2d.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import bench as Bn
dev = torch.device("cuda", 0)
flow, prior, _, _ = Bn.build_models(3, 16, "bf16", dev)
u8, noise, u8_d, noise_d = Bn.make_batch(512, 0, dev)
u8p, noisep = u8.pin_memory(), noise.pin_memory()
with torch.no_grad():
    for _ in range(3):
        Bn.step(flow, prior, u8_d, noise_d)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = Bn.step(flow, prior, u8_d, noise_d)

    def timed(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def resident():
        g.replay()

    def with_u8():
        u8_d.copy_(u8p, non_blocking=True)
        g.replay()

    def with_u8_noise():
        u8_d.copy_(u8p, non_blocking=True)
        noise_d.copy_(noisep, non_blocking=True)
        g.replay()

    a, b, c = timed(resident), timed(with_u8), timed(with_u8_noise)
    print(f"batch resident in HBM: {a:.3f} ms/step = {512 / a * 1e3:.0f} img/s")
    print(f"+ H2D of the uint8 images (1.5 MB, pinned): {b:.3f} ms/step = {512 / b * 1e3:.0f} img/s")
    print(f"+ H2D of images and fp32 dequantisation noise (7.5 MB): {c:.3f} ms/step = {512 / c * 1e3:.0f} img/s")
