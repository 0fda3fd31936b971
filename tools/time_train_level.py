#!/usr/bin/env python3
"""Per-level time of the TRAINING forward launch (nfdp_level_fused_train_bf16) against the inference launch (nfdp_level_fused_bf16)
on the BASELINE shape: Glow L3 K16, B images.  usage: python tools/time_train_level.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import Glow
from normalizing_flow.glow import _level_tables

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
flow = Glow(in_channel=3, L=3, K=16, compute="bf16")
flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 1, DEEP))) for k, v in flow.state_dict().items()})
flow.to(dev)
levels = [list(b.flows) for b in flow.blocks] + [list(flow.final_flows)]
shapes = [(16, 12), (8, 24), (4, 48)]


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


with torch.no_grad():
    for flows, (hw, c) in zip(levels, shapes):
        x = torch.randn(B, hw, hw, c, device=dev) * 0.3
        ldj = torch.zeros(B, dtype=torch.float64, device=dev)
        tabs = _level_tables(flows, x)
        ws = [f.invconv2d._w() for f in flows]
        packed_t = [f.affcoupling.net.fused_packed_train() for f in flows]
        t_inf = timed(lambda: K.level_fused(x.clone(), tabs[0], tabs[1], tabs[2], ws, ldj, inverse=False))
        t_clone = timed(lambda: x.clone())
        keep = []
        def train():
            keep.clear()
            keep.append(K.level_fused_train(x.clone(), packed_t, tabs[1], tabs[2], ws, ldj))
        t_tr = timed(train)
        gb = 2 * len(flows) * B * hw * hw * 512 * 2 / 1e9
        print(f"level {hw}x{hw}x{c}: inference {t_inf - t_clone:8.1f} us   training {t_tr - t_clone:8.1f} us   "
              f"(+{gb:.2f} GB hidden stores -> {gb / ((t_tr - t_clone) * 1e-6) / 1e3:.2f} TB/s if it were the only cost)", flush=True)
