#!/usr/bin/env python3
"""Phase stamps of conv3x3_ldsx_kernel with and without the input GroupNorm (build: tools/build_variant_tu.sh lxs conv.hip -DNFDP_LX_STAMPS;
NFDP_LIB_PATH=.../libnfdp_hip_lxs.so): start, [tables], staging issued, staged (barrier), main loop, barrier, end; medians over the
sampled workgroups."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
dev = "cuda"
for (b, hw, c) in [(64, 16, 64), (64, 8, 128), (64, 8, 64)]:
    x = torch.randn(b, hw, hw, c, device=dev)
    w = torch.randn(c, c, 3, 3, device=dev) * 0.05
    packed = K.conv_pack(w, K.F32)
    ga, be = torch.randn(c, device=dev), torch.randn(c, device=dev)
    ss = torch.randn(b, 2 * c, device=dev) * 0.2
    raw, sums, tile = K.conv2d_stats(x, packed, c, 3, None, 8)
    for mode in ("plain", "gn-in"):
        run = (lambda: K.conv2d(raw, packed, c, 3)) if mode == "plain" else (lambda: K.conv2d_gnin(raw, sums, tile, 8, ga, be, ss, packed, c, None))
        buf = torch.zeros(64 * 16, dtype=torch.int64, device=dev)
        for _ in range(3):
            run()
        K.lib().nfdp_debug_set_conv_stamps(ctypes.c_void_p(buf.data_ptr()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        K.lib().nfdp_debug_set_conv_stamps(ctypes.c_void_p(0))
        t = buf.cpu().numpy().reshape(64, 8, 2)
        t = t[t[:, 0, 0] != 0]
        order = [0, 6, 1, 2, 3, 4, 5]
        ns = np.diff(t[:, order, 1], axis=1) * 10
        names = ["first loads+tables", "staging", "stage barrier", "main loop", "barrier", "epilogue"]
        print(f"B={b} {hw}x{hw} {c}->{c} {mode}: " + "  ".join(f"{n} {int(np.median(ns[:, i]))}" for i, n in enumerate(names)) +
              f"  | total {int(np.median((t[:, 5, 1] - t[:, 0, 1]) * 10))} ns")
