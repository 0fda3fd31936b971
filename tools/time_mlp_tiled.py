#!/usr/bin/env python3
"""HIP-event time of the tiled coupling-network kernels alone (nfdp_coupling_mlp_gather_bf16 / _bf16x3), per geometry.
usage: python tools/time_mlp_tiled.py   (NFDP_LIB_PATH selects an ablation build: tools/build_variant_tu.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import AffineCoupling
dev = torch.device("cuda", 0)
for c, hw, b in ((12, 64, 32), (24, 32, 32), (48, 16, 32), (12, 16, 512), (24, 8, 512), (48, 4, 512), (24, 8, 64), (12, 16, 64)):
    m = AffineCoupling(c, n_features=512, compute="bf16")
    m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor("a." + k, v.shape, 3, DEEP))) for k, v in m.state_dict().items()})
    m.to(dev)
    xh = (torch.rand(b, hw, hw, c, device=dev) - 0.5)
    zc = m.net[-1]
    for name, split in (("bf16", False), ("bf16x3", True)):
        if split and c > 24 and c != 48:
            continue
        kind = K.coupling_mlp_image_kind(xh[..., : c // 2], c, split=split)        # 2: the 16x16x32 form (C = 4, 12 beyond the row-split sizes)
        hi = m.net.fused_packed(kind)
        kw = {"packed_lo": m.net.fused_packed(kind + 2)} if split else {}
        fn = lambda: K.coupling_net_tiled(xh[..., : c // 2], hi, c, zc.bias, zc.logs.view(-1), zc.logscale_factor, return_taps=True, kind=kind, **kw)
        for _ in range(3):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(20):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / 20 * 1e3
        fl = 2.0 * b * hw * hw * (9 * (c // 2) * 512 + 512 * 512 + 9 * 512 * c) * (3 if name == "bf16x3" else 1)
        print(f"C={c:3d} {hw}x{hw} B={b:3d} {name:7s} kind {kind}: {us:8.1f} us  {fl / us / 1e6:7.0f} TFLOP/s of bf16 MFMA", flush=True)
