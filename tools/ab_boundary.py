#!/usr/bin/env python3
"""Same-box A/B of glow.FUSED_LEVEL_BOUNDARY on the headline step (Glow L3 K16, B = 512, bf16, hipGraph replay): median of 200
HIP-event-timed replays per setting, alternating captures.  usage: python3 tools/ab_boundary.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import bench
from normalizing_flow import glow as G
dev = torch.device("cuda", 0)
flow, prior, _, _ = bench.build_models(3, 16, "bf16", dev)
_, _, u8_d, noise_d = bench.make_batch(512, 0, dev)
res = {}
side_levels = int(os.environ.get("AB_SIDE_LEVELS", "0"))
with torch.no_grad():
    for rnd in range(2):
        for fused in (True, False):
            if side_levels:                                  # A/B of the log-det side branch for 3-level models instead
                G.LOGDETS_SIDE_MIN_LEVELS = 3 if fused else 4
            else:
                G.FUSED_LEVEL_BOUNDARY = fused
            for _ in range(2):
                bench.step(flow, prior, u8_d, noise_d)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = bench.step(flow, prior, u8_d, noise_d)
            for _ in range(60):
                g.replay()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
            for e0, e1 in ev:
                e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ev)
            res.setdefault(fused, []).append(ms[100])
            print(f"round {rnd} fused boundary {fused}: median {ms[100]:.4f} ms  min {ms[0]:.4f}  bits/dim {float(out[0] / out[1]):.6f}", flush=True)
print(f"fused {min(res[True]):.4f} ms vs per-op {min(res[False]):.4f} ms")
