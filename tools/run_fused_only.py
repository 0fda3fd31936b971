#!/usr/bin/env python3
"""Profiling driver: runs only the fused coupling-layer kernel at the three north-star geometries (B=512) so that
rocprofv3 --pmc passes see clean per-kernel counters.  usage: python tools/run_fused_only.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np
import torch
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import StepFlow

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
B = 512
with torch.no_grad():
    for c, hw in ((12, 16), (24, 8), (48, 4)):
        m = StepFlow(c, 512, compute="bf16")
        m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor("flows.0." + k, v.shape, 1, DEEP)))
                           for k, v in m.state_dict().items()})
        x = (torch.rand(B, hw, hw, c, device=dev) - 0.5)
        ldj = torch.zeros(B, dtype=torch.float64, device=dev)
        m.transform_nhwc_(x, ldj, None, defer_logdet=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            m.transform_nhwc_(x, ldj, None, defer_logdet=True)
        e1.record()
        torch.cuda.synchronize()
        print(f"C={c} hw={hw}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch")
print("done")
