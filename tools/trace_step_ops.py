#!/usr/bin/env python3
"""Everything one eager headline step (bench.step: forward + NLL, L3 K16 B=512 bf16) puts on the device, in stream order, with the
gaps between launches: torch.profiler device rows (kernels, memcpys, memsets)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import bench
dev = torch.device("cuda", 0)
flow, prior, _, _ = bench.build_models(3, 16, "bf16", dev)
_, _, u8_d, noise_d = bench.make_batch(512, 0, dev)
from normalizing_flow import glow
glow.set_weight_term_cache(False)
with torch.no_grad():
    for _ in range(3):
        bench.step(flow, prior, u8_d, noise_d)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        bench.step(flow, prior, u8_d, noise_d)
        torch.cuda.synchronize()
    # the same step as a graph replay
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        bench.step(flow, prior, u8_d, noise_d)
    g.replay(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof2:
        g.replay()
        torch.cuda.synchronize()
for name, pr in (("eager", prof), ("graph replay", prof2)):
    ev = [e for e in pr.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ev.sort(key=lambda e: e.time_range.start)
    t0 = ev[0].time_range.start
    print(f"--- {name}: {len(ev)} device activities, span {(ev[-1].time_range.end - t0):.1f} us, busy {sum(e.time_range.end - e.time_range.start for e in ev):.1f} us")
    prev = t0
    for e in ev:
        print(f"{e.time_range.start - t0:9.1f} us  +gap {e.time_range.start - prev:6.1f}  dur {e.time_range.end - e.time_range.start:8.1f}  {e.name[:90]}")
        prev = e.time_range.end
