import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
for c in (12, 24, 48, 96, 192, 4, 8, 16):
    w = (torch.linalg.qr(torch.randn(16, c, c))[0] + 0.1 * torch.randn(16, c, c)).contiguous().cuda()
    lad, winv = K.invconv_prepare(w, True)
    ref = torch.linalg.inv(w.double())
    sl = torch.linalg.slogdet(w.double())[1]
    e1 = float((winv.double() - ref).norm() / ref.norm()); e2 = float((lad.double() - sl).abs().max())
    for _ in range(3): K.invconv_prepare(w, True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); K.invconv_prepare(w, True); b.record()
    torch.cuda.synchronize()
    print(f"C={c}: inverse rel err {e1:.2e}, logdet abs err {e2:.2e}, {np.median([a.elapsed_time(b) for a, b in ev]) * 1e3:.1f} us per call", flush=True)
