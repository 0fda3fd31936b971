#!/usr/bin/env python3
"""Phase stamps of conv3x3_ldsx_kernel (-DNFDP_DIAG -DNFDP_LX_STAMPS build: tools/build_variant_tu.sh lxs conv.hip -DNFDP_LX_STAMPS;
NFDP_LIB_PATH=.../libnfdp_hip_lxs.so): wave 0 of every 37th workgroup stores s_memtime / s_memrealtime at start, staging issued, staged
(barrier), main loop done, barrier, end.  Printed: shader cycles and ns between the stamps, median over the sampled workgroups."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
dev = "cuda"
for (b, hw, cin, cout) in [(64, 16, 64, 64), (64, 16, 128, 64), (64, 8, 128, 128), (64, 8, 192, 128), (64, 4, 128, 128)]:
    x = torch.randn(b, hw, hw, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    packed = K.conv_pack(w, K.F32)
    out = torch.empty(b, hw, hw, cout, device=dev)
    buf = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
    for _ in range(3):
        K.conv2d(x, packed, cout, 3, out=out)
    K.lib().nfdp_debug_set_conv_stamps(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        K.conv2d(x, packed, cout, 3, out=out)
    torch.cuda.synchronize()
    K.lib().nfdp_debug_set_conv_stamps(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(64, 8, 2)
    t = t[t[:, 0, 0] != 0]
    cyc = np.diff(t[:, :6, 0], axis=1)
    ns = np.diff(t[:, :6, 1], axis=1) * 10
    span = (t[:, 5, 1].max() - t[:, 0, 1].min()) * 10
    names = ["stage issue", "stage barrier", "main loop", "barrier", "epilogue"]
    print(f"B={b} {hw}x{hw} {cin}->{cout}: {len(t)} workgroups sampled, first start to last end {span} ns")
    for i, n in enumerate(names):
        print(f"    {n:14s} {int(np.median(cyc[:, i])):7d} cycles  {int(np.median(ns[:, i])):6d} ns   (min {int(ns[:, i].min())}, max {int(ns[:, i].max())})")
