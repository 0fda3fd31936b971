#!/usr/bin/env python3
"""Latency of the small fp32 convs of the U-Net reverse step (HIP events over back-to-back launches).
usage: python tools/bench_small_conv.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import nfdp_hip as K
dev = "cuda"
for (b, hw, cin, cout, k) in [(16, 16, 64, 64, 3), (64, 16, 64, 64, 3), (128, 16, 64, 64, 3), (256, 16, 64, 64, 3), (1024, 16, 64, 64, 3),
                              (64, 16, 128, 64, 3), (64, 8, 64, 64, 3), (64, 8, 128, 128, 3), (64, 8, 192, 128, 3), (64, 4, 128, 128, 3), (64, 16, 128, 64, 1)]:
    x = torch.randn(b, hw, hw, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed = K.conv_pack(w, K.F32)
    out = torch.empty(b, hw, hw, cout, device=dev)
    for _ in range(5):
        K.conv2d(x, packed, cout, k, out=out)
    n = 100
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        K.conv2d(x, packed, cout, k, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * b * hw * hw * cin * k * k * cout
    print(f"B={b:5d} {hw}x{hw} {cin}->{cout} k{k}: {us:8.1f} us/launch  {fl / us / 1e6:7.1f} TFLOP/s  ({-(-b * hw * hw // 64) * -(-cout // 64)} tiles)", flush=True)
