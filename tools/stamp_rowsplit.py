#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime stamps of the row-split (hidden tiles in LDS) fused kernel at the 8x8 level (needs a
-DNFDP_FUSED_STAMPS build of the library, path in NFDP_LIB_PATH).  Median cycles per phase over the workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import StepFlow
names = ["load X", "head", "im2col", "GEMM1", "GEMM2 loop", "H2 exchange", "GEMM3 loop", "T write", "shift-sum+coupling", "store"]
dev = torch.device("cuda", 0)
with torch.no_grad():
    for c, hw, B in [(24, 8, b) for b in (int(v) for v in (sys.argv[1:] or ["512"]))]:
        flows = []
        for f in range(4):                                   # different weights per launch, like consecutive flows
            m = StepFlow(c, 512, compute="bf16")
            m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(f"flows.{f}." + k, v.shape, 1, DEEP))) for k, v in m.state_dict().items()})
            flows.append(m)
        x = torch.rand(B, hw, hw, c, device=dev) - 0.5
        ldj = torch.zeros(B, dtype=torch.float64, device=dev)
        for _ in range(3):
            for m in flows:
                m.transform_nhwc_(x, ldj, None, defer_logdet=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for m in flows:
            m.transform_nhwc_(x, ldj, None, defer_logdet=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / len(flows)
        buf = (ctypes.c_ulonglong * (512 * 16))()
        f = getattr(ctypes.CDLL(nfdp_hip.LIB_PATH), "nfdp_debug_read_stamps_a")
        f.argtypes = [ctypes.c_void_p]
        assert f(ctypes.addressof(buf)) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)
        nwg = B // 2
        d = np.diff(a[:nwg, :11], axis=1)
        med = np.median(d, axis=0)
        tot = np.median(a[:nwg, 10] - a[:nwg, 0])
        xs = [w for w in range(nwg) if w % 8 == 0]
        span = a[xs, 10].max() - a[xs, 0].min()                # one XCD's workgroups share a counter
        print(f"C={c} hw={hw} B={B} ({nwg} WGs): median total {tot:.0f} cyc/WG, XCD-0 span {span} ticks, {us:.1f} us/launch by events -> {span / us / 1e3:.2f} ticks/ns; " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, med)))
