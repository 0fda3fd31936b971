#!/usr/bin/env python3
"""The floor of a dependent kernel chain inside a hipGraph on this box: 200 back-to-back launches of a trivial kernel (a 1 K-element fill,
a 4 MB elementwise add) and of the U-Net's small convs, per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import nfdp_hip as K
dev = "cuda"


def per_launch(fn, n=200):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(5):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3


t = torch.zeros(1024, device=dev)
a = torch.randn(64, 16, 16, 64, device=dev)
print(f"torch fill_ of 1 K floats:            {per_launch(lambda: t.fill_(1.0)):6.2f} us per launch")
print(f"torch add_ on 4 MB (64x16x16x64 f32): {per_launch(lambda: a.add_(1.0)):6.2f} us per launch")
x = torch.randn(64, 16, 16, 64, device=dev)
for cin, cout, k in ((64, 64, 1), (64, 64, 3), (128, 64, 3)):
    xx = torch.randn(64, 16, 16, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed = K.conv_pack(w, K.F32)
    out = torch.empty(64, 16, 16, cout, device=dev)
    print(f"conv k{k} {cin}->{cout} 64x16x16 (fp32, split products): {per_launch(lambda: K.conv2d(xx, packed, cout, k, out=out)):6.2f} us per launch")
gam, bet = torch.ones(64, device=dev), torch.zeros(64, device=dev)
print(f"groupnorm_silu 64x16x16x64 g8:        {per_launch(lambda: K.groupnorm_silu(x, 8, gam, bet, None)):6.2f} us per launch")
g64 = torch.ones(64, device=dev)
print(f"chan_layernorm 64x16x16x64:           {per_launch(lambda: K.chan_layernorm(x, g64)):6.2f} us per launch")
