#!/usr/bin/env python3
"""Per-kernel HIP-event trace of one Glow forward on the per-op path: where the time goes.
usage: python tools/trace_glow_fwd.py [size=32] [L=3] [B=512] [compute=f32] [f32_coupling=bf16x3]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import Glow
arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d
size, L, B, compute, mode = arg(1, 32), arg(2, 3), arg(3, 512), arg(4, "f32"), arg(5, "bf16x3")
dev = torch.device("cuda", 0)
K.set_f32_coupling(mode)
with torch.no_grad():
    flow = Glow(in_channel=3, L=L, K=16, compute=compute)
    flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 1, DEEP))) for k, v in flow.state_dict().items()})
    flow.to(dev)
    xh = (torch.rand(B, size, size, 3, device=dev) - 0.5)
    z = lambda: torch.zeros(B, dtype=torch.float64, device=dev)
    for _ in range(2):
        flow.transform_nhwc(xh, z(), z())
    K.start_trace()
    flow.transform_nhwc(xh, z(), z())
    tr = K.stop_trace()
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for e in tr:
    a = agg[e["label"]]
    a[0] += e["ms"]; a[1] += 1; a[2] += e["flops"]
tot = sum(v[0] for v in agg.values())
print(f"{size}x{size} L{L} K16 B={B} {compute} ({mode}): {len(tr)} launches, {tot:.2f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(arg(6, 18))]:
    tf = f"  {v[2] / v[0] / 1e9:.0f} TFLOP/s" if v[2] else ""
    print(f"  {v[0]:8.3f} ms  x{v[1]:3d}  {k}{tf}")
