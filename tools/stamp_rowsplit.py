#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime stamps of the row-split flow-step kernels (8x8: coupling_fused_rs2_kernel, 4x4: coupling_fused_rs_kernel).
Needs a -DNFDP_FUSED_STAMPS -DNFDP_FUSED_STAMPS_FINE build (tools/build_variant.sh stamps ...), selected with NFDP_LIB_PATH.
Prints the median cycles between consecutive stamps over the workgroups of a one-flow launch (the fine stamps perturb these kernels:
read the split, not the total)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import StepFlow
dev = torch.device("cuda", 0)
with torch.no_grad():
    for c, hw, fn, nwg in ((24, 8, "nfdp_debug_read_stamps_v", 256), (48, 4, "nfdp_debug_read_stamps_a", 256)):
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
        a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)[:nwg]
        cols = [k for k in range(11) if np.median(a[:, k]) > 0]
        med = {k: np.median(a[:, k] - a[:, 0]) for k in cols}
        print(f"C={c} hw={hw}: stamps at (cycles since stamp 0, median over {nwg} WGs): " + ", ".join(f"[{k}] {med[k]:.0f}" for k in cols))
        prev = None
        for k in cols:
            if prev is not None:
                print(f"    stamp {prev} -> {k}: {np.median(a[:, k] - a[:, prev]):.0f} cycles")
            prev = k
