"""Achieved HBM bandwidth of the bandwidth-bound flow kernels (ActNorm / inv-1x1 target >= 60 % of HBM peak).

Each op is timed standalone with HIP events over `--iters` back-to-back launches at the three Glow-L3 level shapes
(C = 12/24/48 after the squeeze of a 3x32x32 image) for a batch that fits the caches (B=512: 3-13 MB per tensor, lives in
the 256 MB Infinity Cache) and one that does not (B=16384: 100-400 MB per tensor).  Algorithmic bytes = one read + one
write of the fp32 NHWC tensor (2*C*4 B per pixel), as in DESIGN.md §4.

    python tools/bench_hbm_ops.py [--iters 50] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(K, fn, iters):
    """Median per-launch duration in us from HIP events recorded around each launch (binding's launch trace)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    K.start_trace()
    for _ in range(iters):
        fn()
    ms = sorted(r["ms"] for r in K.stop_trace())
    return ms[len(ms) // 2] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import nfdp_hip as K
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    rows = []
    for B in (512, 16384):
        for C, hw in ((12, 16), (24, 8), (48, 4)):
            npix = B * hw * hw
            x = torch.randn(npix, C, generator=g).to(dev)
            y = torch.empty_like(x)
            scale = (0.1 * torch.randn(C, generator=g)).to(dev)
            bias = (0.1 * torch.randn(C, generator=g)).to(dev)
            w = torch.linalg.qr(torch.randn(C, C, generator=g))[0].contiguous().to(dev)
            ldj = torch.zeros(B, dtype=torch.float64, device=dev)
            nbytes = 2 * npix * C * 4
            ops = {
                "actnorm_fwd": (lambda: K.actnorm_fwd(x, scale, bias, out=y), nbytes),
                "actnorm_inv": (lambda: K.actnorm_inv(x, scale, bias, out=y), nbytes),
                "invconv_apply": (lambda: K.invconv_apply(x, w, out=y), nbytes),
                "actnorm_invconv_fwd (flow head)": (lambda: K.actnorm_invconv_fwd(x, scale, bias, w, out=y), nbytes),
                "invconv_actnorm_inv": (lambda: K.invconv_actnorm_inv(x, scale, bias, w, out=y), nbytes),
            }
            # Squeeze as Glow calls it at this level: input = the kept half of the previous Split (C/4 channels at
            # pixel stride C/2; the 3-channel image itself at level 0), output = this level's [B,hw,hw,C] tensor.
            cin = C // 4
            xsrc = torch.randn(B, 2 * hw, 2 * hw, cin if C == 12 else 2 * cin, generator=g).to(dev)[..., :cin]
            ops["squeeze (level input)"] = (lambda: K.squeeze(xsrc), nbytes)
            ys = x.view(B, hw, hw, C)
            ops["unsqueeze"] = (lambda: K.unsqueeze(ys), nbytes)
            for name, (fn, nb) in ops.items():
                try:
                    us = timed(K, fn, args.iters)
                except Exception as e:                               # noqa: BLE001
                    print(f"{name} B={B} C={C}: {type(e).__name__}: {e}", file=sys.stderr)
                    continue
                gbs = nb / us * 1e-3
                rows.append({"op": name, "B": B, "C": C, "HW": hw * hw, "MB": round(nb / 1e6, 2), "us": round(us, 2),
                             "GB/s": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 3)})
                print(f"{name:34s} B={B:6d} C={C:2d} {nb / 1e6:8.1f} MB  {us:9.2f} us  {gbs:8.1f} GB/s  "
                      f"{100 * gbs / HBM_PEAK_GBS:5.1f} % of 8 TB/s", flush=True)
            del x, y
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"hbm_peak_GBs": HBM_PEAK_GBS, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
