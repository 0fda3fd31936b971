#!/usr/bin/env python3
"""One weight-gradient GEMM of the coupling network in isolation (for PMC passes and A/B timing).
usage: python tools/bench_wgrad.py [which=dw2|dw1|dw3] [hw=16] [c=12] [B=512] [reps=10]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K

which = sys.argv[1] if len(sys.argv) > 1 else "dw2"
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 16
c = int(sys.argv[3]) if len(sys.argv) > 3 else 12
B = int(sys.argv[4]) if len(sys.argv) > 4 else 512
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda", 0)
grp = {16: 1, 8: 2, 4: 8}[hw]
bpad = (B + grp - 1) // grp * grp
npx = bpad * hw * hw
g = torch.Generator(device=dev).manual_seed(1)
hid = lambda: torch.randn(bpad, hw, hw, 512, device=dev, generator=g).to(torch.bfloat16)
gs, gb = torch.zeros(512, device=dev), torch.zeros(512, device=dev)
if which == "dw2":
    y1, dc2 = torch.relu(hid()), hid()
    w2, b2 = torch.randn(512, 512, 1, 1, device=dev) * 0.05, torch.randn(512, device=dev) * 0.1
    gw = torch.zeros_like(w2)
    fn = lambda: K.conv_wgrad_affine(y1, dc2, w2, b2, gw, gs, gb, True, x_blocked=True, dy_blocked=True)
    flops = 2.0 * npx * 512 * 512
elif which == "dw1":
    xcol = K.im2col3x3(torch.randn(B, hw, hw, c // 2, device=dev), pad_batch=bpad)
    dc1 = hid()
    w1, b1 = torch.randn(512, c // 2, 3, 3, device=dev) * 0.05, torch.randn(512, device=dev) * 0.1
    gw = torch.zeros_like(w1)
    fn = lambda: K.conv_wgrad_affine(xcol, dc1, w1, b1, gw, gs, gb, True, form=1, form_c=c // 2, dy_blocked=True)
    flops = 2.0 * npx * 512 * xcol.shape[-1]
else:
    dcol = K.im2col3x3(torch.randn(B, hw, hw, c, device=dev), pad_batch=bpad)
    y2 = torch.relu(hid())
    gw = torch.zeros(c, 512, 3, 3, device=dev)
    fn = lambda: K.conv_wgrad_blocked(dcol, y2, gw, True, form=2, form_c=c, dy_blocked=True)
    flops = 2.0 * npx * 512 * dcol.shape[-1]
for _ in range(3):
    fn()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for e0, e1 in ev:
    e0.record(); fn(); e1.record()
torch.cuda.synchronize()
us = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
print(f"{which} {hw}x{hw} C={c} B={B}: {us:.1f} us per call (GEMM + reduce), {flops / us / 1e6:.0f} TFLOP/s", flush=True)
