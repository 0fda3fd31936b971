#!/usr/bin/env python3
"""The small fp32 3x3 convs of the U-Net reverse step in the three product modes (exact fp32 MFMA chain, six bf16 MFMAs on three-way splits,
plain bf16 operands) as back-to-back launches inside a hipGraph: what the split arithmetic costs beyond the kernel's own floor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import nfdp_hip as K
dev = "cuda"
for (b, hw, cin, cout, k) in [(64, 16, 64, 64, 3), (64, 16, 128, 64, 3), (64, 8, 128, 128, 3), (64, 8, 192, 128, 3), (64, 4, 128, 128, 3)]:
    x = torch.randn(b, hw, hw, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    row = []
    for mode, comp in (("exact", K.F32), ("bf16x6", K.F32), ("bf16", K.BF16)):
        if comp == K.F32:
            K.set_f32_products(mode)
        packed = K.conv_pack(w, comp)
        out = torch.empty(b, hw, hw, cout, device=dev)
        for _ in range(3):
            K.conv2d(x, packed, cout, k, out=out, compute=comp)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50):
                K.conv2d(x, packed, cout, k, out=out, compute=comp)
        for _ in range(5):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record()
        torch.cuda.synchronize()
        row.append(f"{mode} {e0.elapsed_time(e1) / 100 * 1e3:6.1f} us")
    K.set_f32_products("bf16x6")
    print(f"B={b} {hw}x{hw} {cin}->{cout} k{k}:  " + "   ".join(row), flush=True)
