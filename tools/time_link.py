import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
dev = "cuda"
for c, hw in [(12, 16), (24, 8), (48, 4)]:
    B = 512
    T4 = lambda: torch.randn(B, hw, hw, c, device=dev)
    dx_in, xpre, ypost, hout = T4(), T4(), T4(), T4() * 0.5
    w = torch.linalg.qr(torch.randn(c, c))[0].contiguous().to(dev)
    scale, bias, l3 = torch.randn(c, device=dev) * 0.2, torch.randn(c, device=dev) * 0.3, torch.randn(c, device=dev) * 0.1
    gl = torch.randn(B, device=dev, dtype=torch.float64)
    npix = B * hw * hw
    part = K.flow_link_part(1, npix, c, dev)
    fn = lambda: K.flow_link_bwd(dx_in, part[0], head=(xpre, w, scale, bias), coupling=(ypost, hout, gl, l3, 3.0), want_a=False)
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    print(f"link C={c} hw={hw}: {np.median([a.elapsed_time(b) for a, b in ev]) * 1e3:.1f} us", flush=True)
