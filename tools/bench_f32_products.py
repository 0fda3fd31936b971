#!/usr/bin/env python3
"""fp32 products on the bf16 matrix cores (nfdp_set_f32_products): accuracy against fp64 and latency of the small fp32 convs,
exact v_mfma_f32_16x16x4_f32 chain vs three-way bf16 split + six bf16 MFMAs.  usage: python tools/bench_f32_products.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import torch.nn.functional as F
import nfdp_hip as K
dev = "cuda"
for (b, hw, cin, cout, k) in [(64, 16, 64, 64, 3), (64, 8, 128, 128, 3), (64, 8, 192, 128, 3), (16, 8, 512, 64, 3),
                              (32, 64, 64, 64, 3), (32, 64, 128, 64, 3), (512, 16, 512, 512, 1), (512, 16, 6, 512, 3)]:
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(b, hw, hw, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=k // 2).permute(0, 2, 3, 1)
    packed = K.conv_pack(w.to(dev), K.F32)
    xd = x.to(dev)
    out = torch.empty(b, hw, hw, cout, device=dev)
    line = f"B={b} {hw}x{hw} {cin}->{cout} k{k} (K={cin * k * k}):"
    for mode in ("exact", "bf16x6"):
        K.set_f32_products(mode)
        for _ in range(5):
            K.conv2d(xd, packed, cout, k, out=out)
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            K.conv2d(xd, packed, cout, k, out=out)
        e1.record()
        torch.cuda.synchronize()
        line += f"  {mode}: {e0.elapsed_time(e1) / n * 1e3:6.1f} us, err vs fp64 {err:.1e};"
    print(line, flush=True)
K.set_f32_products("exact")
