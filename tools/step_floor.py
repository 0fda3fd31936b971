#!/usr/bin/env python3
"""What the headline step (Glow L3 K16, B = 512, bf16, hipGraph replay) costs beyond its three level launches: the whole step against a
graph of ONLY the three level launches (in place on buffers of the right shapes), both as medians of 200 event-timed replays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import bench
import nfdp_hip as K
from normalizing_flow.glow import _run_flows_fwd
dev = torch.device("cuda", 0)
flow, prior, _, _ = bench.build_models(3, 16, "bf16", dev)
_, _, u8_d, noise_d = bench.make_batch(512, 0, dev)


def med(g, n=200):
    for _ in range(5):
        g.replay()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


with torch.no_grad():
    for _ in range(2):
        bench.step(flow, prior, u8_d, noise_d)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        bench.step(flow, prior, u8_d, noise_d)
    t_step = med(g)
    xh = K.preprocess_u8(u8_d, noise_d, 5)
    bufs = [K.squeeze(xh)]
    for i in range(2):
        bufs.append(K.squeeze(bufs[-1][..., : bufs[-1].shape[-1] // 2].contiguous()))
    levels = [list(b.flows) for b in flow.blocks] + [list(flow.final_flows)]
    ldj = torch.zeros(512, dtype=torch.float64, device=dev)
    for fl, bf in zip(levels, bufs):
        _run_flows_fwd(fl, bf, ldj, None, defer_logdet=True)
    torch.cuda.synchronize()
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        for fl, bf in zip(levels, bufs):
            _run_flows_fwd(fl, bf, ldj, None, defer_logdet=True)
    t3 = med(g3)
    each = []
    for fl, bf in zip(levels, bufs):
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            _run_flows_fwd(fl, bf, ldj, None, defer_logdet=True)
        each.append(med(g1))
print(f"whole step {t_step:.4f} ms; three level launches alone {t3:.4f} ms (one at a time: {' + '.join(f'{e:.4f}' for e in each)} = {sum(each):.4f}); "
      f"everything else {1e3 * (t_step - t3):.0f} us")
