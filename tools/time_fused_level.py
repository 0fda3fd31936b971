#!/usr/bin/env python3
"""Per-launch time of the fused flow-step kernel at one Glow level, over 16 flows with distinct weights (as in a real
forward pass, so weights are not L2-warm from the previous launch).  HIP events around every launch; median reported.
usage: python tools/time_fused_level.py [C hw]...   (default: all three levels of the 3x32x32 model), B=512."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import StepFlow

args = [int(v) for v in sys.argv[1:]]
levels = list(zip(args[0::2], args[1::2])) or [(12, 16), (24, 8), (48, 4)]
B = int(os.environ.get("NFDP_B", "512"))
dev = torch.device("cuda", 0)
with torch.no_grad():
    for c, hw in levels:
        flows = []
        for f in range(16):
            m = StepFlow(c, 512, compute="bf16")
            m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(f"flows.{f}." + k, v.shape, 1, DEEP)))
                               for k, v in m.state_dict().items()})
            flows.append(m.to(dev))
        x = torch.rand(B, hw, hw, c, device=dev) - 0.5
        ldj = torch.zeros(B, dtype=torch.float64, device=dev)
        if not os.environ.get("NFDP_LEVEL_ONLY"):            # per-flow launches (one kernel per flow step)
            for _ in range(2):
                for m in flows:
                    m.transform_nhwc_(x, ldj, None, defer_logdet=True)
            torch.cuda.synchronize()
            K.start_trace()
            for _ in range(3):
                for m in flows:
                    m.transform_nhwc_(x, ldj, None, defer_logdet=True)
            recs = [r for r in K.stop_trace() if "fused" in r["label"]]
            us = sorted(r["ms"] * 1e3 for r in recs)
            print(f"C={c} hw={hw} B={B}: {len(us)} launches, median {us[len(us) // 2]:.1f} us, min {us[0]:.1f}, max {us[-1]:.1f}", flush=True)
        if hasattr(K, "level_fused"):                       # the same 16 flows as ONE persistent launch
            from normalizing_flow.glow import _level_tables
            packed, scales, biases = _level_tables(flows, x)
            ws = [f.invconv2d._w() for f in flows]
            if os.environ.get("NFDP_SAME_W"):                 # every flow reads flow 0's image: the weights stay L2-resident
                packed = [packed[0]] * len(packed)
            for _ in range(2):
                K.level_fused(x, packed, scales, biases, ws, ldj, inverse=False)
            torch.cuda.synchronize()
            K.start_trace()
            for _ in range(5):
                K.level_fused(x, packed, scales, biases, ws, ldj, inverse=False)
            lus = sorted(r["ms"] * 1e3 for r in K.stop_trace())
            print(f"   level launch (16 flows): median {lus[len(lus) // 2]:.1f} us = {lus[len(lus) // 2] / 16:.1f} us per flow", flush=True)
        if os.environ.get("NFDP_READ_STAMPS"):                 # -DNFDP_FUSED_STAMPS build: start/end s_memtime of the last launch
            import ctypes
            buf = (ctypes.c_ulonglong * (512 * 16))()
            f = getattr(ctypes.CDLL(K.LIB_PATH), os.environ["NFDP_READ_STAMPS"])
            f.argtypes = [ctypes.c_void_p]
            assert f(ctypes.addressof(buf)) == 0
            a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)
            nwg = min(512, B * hw * hw // 128 if hw == 8 else B * hw * hw // 32)
            d = a[:nwg, 10] - a[:nwg, 0]
            print(f"   s_memtime ticks per workgroup: median {np.median(d):.0f}, max {d.max()} -> {np.median(d) / us[len(us) // 2] / 1e3:.2f} ticks/ns "
                  f"if a workgroup spans the launch", flush=True)
