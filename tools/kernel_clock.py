#!/usr/bin/env python3
"""In-kernel shader clock of the level-0 fused flow-step launch (16 flows, B=512) on random data.
Needs the diagnostic build:  tools/build_variant.sh clk -DNFDP_FUSED_STAMPS -DNFDP_FUSED_CLOCK
then  NFDP_LIB_PATH=.../libnfdp_hip_clk.so python tools/kernel_clock.py
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup, after ~2 s of back-to-back launches (the chip settles)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import StepFlow
from normalizing_flow.glow import _level_tables

B = int(os.environ.get("NFDP_B", "512"))
dev = torch.device("cuda", 0)
lib = ctypes.CDLL(K.LIB_PATH)
LEVELS = {"0": (12, 16, "nfdp_debug_read_stamps_v", 8), "1": (24, 8, "nfdp_debug_read_stamps_a", 10), "2": (48, 4, "nfdp_debug_read_stamps_a", 10)}
for c, hw, fn, END in [LEVELS[k] for k in os.environ.get("NFDP_LEVELS", "0").split(",")]:
    with torch.no_grad():
        flows = []
        for f in range(16):
            m = StepFlow(c, 512, compute="bf16")
            m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(f"flows.{f}." + k, v.shape, 1, DEEP)))
                               for k, v in m.state_dict().items()})
            flows.append(m.to(dev))
        x0 = torch.rand(B, hw, hw, c, device=dev) - 0.5
        x = x0.clone()
        ldj = torch.zeros(B, dtype=torch.float64, device=dev)
        packed, scales, biases = _level_tables(flows, x)
        ws = [f.invconv2d._w() for f in flows]
        n = int(os.environ.get("NFDP_LAUNCHES", "1600"))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(n):
            if i == n - 20:
                e0.record()
            x.copy_(x0)
            K.level_fused(x, packed, scales, biases, ws, ldj, inverse=False)
        e1.record()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (512 * 16))()
        f = getattr(lib, fn)
        f.argtypes = [ctypes.c_void_p]
        assert f(ctypes.addressof(buf)) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)
        if os.environ.get("NFDP_PHASES"):       # full -DNFDP_FUSED_STAMPS build: stamps 1..7 are those of the LAST flow of the launch
            a = a[(a[:, 9] > 0) & (a[:, 7] > a[:, 9])]                  # the workgroups this launch stamped
            names = ["head", "im2col", "GEMM1", "GEMM2/3", "tap sums", "coupling+ldj"]
            d = np.diff(a[:, [9, 2, 3, 4, 5, 6, 7]], axis=1)
            print("   last flow, cycles per phase (median over workgroups): " +
                  ", ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(names)) + f"; flow {np.median(a[:, 7] - a[:, 9]):.0f}", flush=True)
            continue
        a = a[(a[:, END] > a[:, 0]) & (a[:, 0] > 0)]
        if END == 10:                       # row-split levels: 512 / 256 workgroups; keep the rows of this launch only
            a = a[: (B * hw * hw) // (64 if hw == 8 and os.environ.get("NFDP_RS2_NCT") != "4" else (128 if hw == 8 else 32))]
        cyc = a[:, END] - a[:, 0]
        rt = a[:, END + 4] - a[:, 4]
        clk = cyc / rt * 0.1            # GHz
        t0 = a[:, 4] - a[:, 4].min()
        print(f"   {len(a)} workgroups; starts (s_memrealtime, 10 ns units after the first): median {np.median(t0):.0f}, max {t0.max()}; "
              f"ends after the first start: median {np.median(a[:, END + 4] - a[:, 4].min()):.0f}, max {(a[:, END + 4] - a[:, 4].min()).max()}", flush=True)
        print(f"C={c} hw={hw} B={B}: launch+copy {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; per workgroup: cycles median {np.median(cyc):.0f}, "
              f"time median {np.median(rt) * 10:.0f} ns, clock median {np.median(clk):.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f})", flush=True)
