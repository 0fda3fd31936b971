#!/usr/bin/env python3
"""Per-kernel HIP-event trace of one Glow L5 K16 128x128x3 forward (B=32, per-op path): where the time goes."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import Glow
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
with torch.no_grad():
    flow = Glow(in_channel=3, L=5, K=16, compute="bf16")
    flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 1, DEEP))) for k, v in flow.state_dict().items()})
    flow.to(dev)
    xh = (torch.rand(B, 128, 128, 3, device=dev) - 0.5)
    for _ in range(2):
        flow.transform_nhwc(xh, torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev))
    K.start_trace()
    flow.transform_nhwc(xh, torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev))
    tr = K.stop_trace()
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for e in tr:
    a = agg[e["label"]]
    a[0] += e["ms"]; a[1] += 1; a[2] += e["flops"]
tot = sum(v[0] for v in agg.values())
print(f"B={B}: {len(tr)} launches, {tot:.2f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:18]:
    tf = f"  {v[2] / v[0] / 1e9:.0f} TFLOP/s" if v[2] else ""
    print(f"  {v[0]:8.2f} ms  x{v[1]:3d}  {k}{tf}")
