#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime stamps of the fused flow-step kernel (needs a -DNFDP_FUSED_STAMPS build of the
library, path in NFDP_LIB_PATH).  Prints median cycles per phase over the workgroups of one launch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import StepFlow
names = ["load X", "head", "im2col", "GEMM1 phase", "GEMM2/3 phase", "tap shift-sum", "coupling+ldj", "(inv head)+store"]
dev = torch.device("cuda", 0)
with torch.no_grad():
    for c, hw, fn in ((12, 16, "nfdp_debug_read_stamps_v"), (24, 8, "nfdp_debug_read_stamps_v"), (48, 4, "nfdp_debug_read_stamps_a")):
        m = StepFlow(c, 512, compute="bf16")
        m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor("flows.0." + k, v.shape, 1, DEEP))) for k, v in m.state_dict().items()})
        x = torch.rand(512, hw, hw, c, device=dev) - 0.5
        ldj = torch.zeros(512, dtype=torch.float64, device=dev)
        for _ in range(3):
            m.transform_nhwc_(x, ldj, None, defer_logdet=True)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (512 * 16))()
        f = getattr(ctypes.CDLL(nfdp_hip.LIB_PATH), fn)
        f.argtypes = [ctypes.c_void_p]
        assert f(ctypes.addressof(buf)) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)
        nwg = {12: 512, 24: 256, 48: 64}[c]
        d = np.diff(a[:nwg, :9], axis=1)
        med = np.median(d, axis=0)
        tot = np.median(a[:nwg, 8] - a[:nwg, 0])
        print(f"C={c} hw={hw}: total {tot:.0f} cyc/WG; " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, med)))
