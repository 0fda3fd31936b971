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
    for c, hw, fn, nwg in ((24, 8, "nfdp_debug_read_stamps_a", 256), (48, 4, "nfdp_debug_read_stamps_a", 256)):
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
        cols = [k for k in range(16) if np.median(a[:, k]) > 0]
        med = {k: np.median(a[:, k] - a[:, 0]) for k in cols}
        cols.sort(key=lambda k: med[k])                      # in time order (stamps 11..: inside the flow head)
        print(f"C={c} hw={hw}: stamps at (cycles since stamp 0, median over {nwg} WGs): " + ", ".join(f"[{k}] {med[k]:.0f}" for k in cols))
        prev = None
        for k in cols:
            if prev is not None:
                print(f"    stamp {prev} -> {k}: {np.median(a[:, k] - a[:, prev]):.0f} cycles")
            prev = k
        # the same 16 times over in ONE level launch (distinct weights per flow): stamps of the LAST flow, relative to its top (stamp 15)
        from normalizing_flow.glow import _level_tables
        flows = []
        for fl in range(16):
            mm = StepFlow(c, 512, compute="bf16")
            mm.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(f"flows.{fl}." + k, v.shape, 1, DEEP))) for k, v in mm.state_dict().items()})
            flows.append(mm.to(dev))
        packed, scales, biases = _level_tables(flows, x)
        ws = [fl.invconv2d._w() for fl in flows]
        for _ in range(3):
            nfdp_hip.level_fused(x, packed, scales, biases, ws, ldj, inverse=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nfdp_hip.level_fused(x, packed, scales, biases, ws, ldj, inverse=False)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        assert f(ctypes.addressof(buf)) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)
        # counters of different XCDs are not aligned: look inside XCD 0 (workgroups 0, 8, 16, ...)
        q = a[0::8]
        q = q[(q[:, 0] > 0) & (q[:, 10] > q[:, 0])]
        st, en = q[:, 0], q[:, 10]
        span = en.max() - st.min()
        late = int((st - st.min() > 0.25 * span).sum())
        print(f"  launch {us:.1f} us by HIP events; XCD 0: first start -> last end {span} ticks = {span / us / 1e3:.2f} ticks/ns; {len(q)} workgroups stamped, "
              f"{late} of them started more than a quarter of the span after the first (a second round: not all resident at once)")
        a = a[:nwg]
        tot = np.median(a[:, 10] - a[:, 0])
        cols = [k for k in range(16) if k != 0 and np.median(a[:, k]) > 0]
        med = {k: np.median(a[:, k] - a[:, 15]) for k in cols}
        cols.sort(key=lambda k: med[k])
        print(f"  level launch of 16 flows: {tot:.0f} cycles per workgroup = {tot / 16:.0f} per flow; last flow, cycles since its top: "
              + ", ".join(f"[{k}] {med[k]:.0f}" for k in cols))
