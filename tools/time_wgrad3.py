#!/usr/bin/env python3
"""The three weight gradients of one flow as one launch (nfdp_coupling_wgrad3_bf16) per Glow level, B=512: time per call (GEMM + reduce)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
dev = "cuda"
B = 512
for c, hw in [(12, 16), (24, 8), (48, 4)]:
    grp = {16: 1, 8: 2, 4: 8}[hw]
    bpad = (B + grp - 1) // grp * grp
    ch = c // 2
    hid = lambda: torch.randn(bpad, hw, hw, 512, device=dev).to(torch.bfloat16)
    y1, y2, dc1, dc2 = torch.relu(hid()), torch.relu(hid()), hid(), hid()
    xcol = K.im2col3x3(torch.randn(B, hw, hw, ch, device=dev), pad_batch=bpad)
    dcol = K.im2col3x3(torch.randn(B, hw, hw, c, device=dev), pad_batch=bpad)
    w1, b1 = torch.randn(512, ch, 3, 3, device=dev) * 0.05, torch.randn(512, device=dev) * 0.1
    w2, b2 = torch.randn(512, 512, 1, 1, device=dev) * 0.05, torch.randn(512, device=dev) * 0.1
    z = lambda *s: torch.zeros(*s, device=dev)
    pr = [dict(x=y1, dy=dc2, dw=z(512, 512, 1, 1), x_blocked=True, weight=w2, bias=b2, dscale=z(512), dbias=z(512)),
          dict(x=xcol, dy=dc1, dw=z(512, ch, 3, 3), form=1, form_c=ch, weight=w1, bias=b1, dscale=z(512), dbias=z(512)),
          dict(x=dcol, dy=y2, dw=z(c, 512, 3, 3), form=2, form_c=c)]
    fn = lambda: K.coupling_wgrad3(pr, True)
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    us = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
    npx = bpad * hw * hw
    fl = 2.0 * npx * 512 * (512 + xcol.shape[-1] + dcol.shape[-1])
    print(f"wgrad3 {hw}x{hw} C={c}: {us:.1f} us per flow ({fl / us / 1e6:.0f} TFLOP/s)", flush=True)
