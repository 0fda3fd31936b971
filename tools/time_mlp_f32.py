#!/usr/bin/env python3
"""Per-launch time of the exact-fp32 on-chip coupling network (nfdp_coupling_mlp_f32) at the three levels of the 3x32x32 model, B = 512,
over 16 distinct weight images (as in a forward pass).  usage: python tools/time_mlp_f32.py [variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import nfdp_hip as K
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(os.environ.get("NFDP_B", "512"))
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for c, hw in [(12, 16), (24, 8), (48, 4)]:
    packs = []
    for f in range(16):
        r = lambda *s: torch.randn(*s, generator=g)
        w1, w2, w3 = r(512, c // 2, 3, 3) / (4.5 * c) ** 0.5, r(512, 512, 1, 1) / 512 ** 0.5, r(c, 512, 3, 3) / 4608 ** 0.5
        s1, b1, s2, b2 = 0.1 * r(512), 0.1 * r(512), 0.1 * r(512), 0.1 * r(512)
        packs.append(K.coupling_mlp_f32_pack(*[t.to(dev).contiguous() for t in (w1, s1, b1, w2.view(512, 512), s2, b2, w3)]))
    x = torch.randn(B, hw, hw, c, generator=g).to(dev)
    bias, logs = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    for _ in range(2):
        for p in packs:
            K.coupling_net_tiled_f32(x[..., : c // 2], p, c, bias, logs, 3.0, variant)
    torch.cuda.synchronize()
    K.start_trace()
    for _ in range(3):
        for p in packs:
            K.coupling_net_tiled_f32(x[..., : c // 2], p, c, bias, logs, 3.0, variant)
    us = sorted(e["ms"] * 1e3 for e in K.stop_trace() if e["label"].startswith("coupling_mlp_f32"))
    fl = 2.0 * B * hw * hw * (9 * (c // 2) * 512 + 512 * 512 + 9 * 512 * c)
    print(f"C={c} hw={hw} B={B}: median {us[len(us) // 2]:.1f} us per flow = {fl / us[len(us) // 2] / 1e6:.1f} TFLOP/s (min {us[0]:.1f})", flush=True)
