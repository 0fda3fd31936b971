import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch, time
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import Glow, glow as G
dev = torch.device("cuda", 0)
B = 32
with torch.no_grad():
    flow = Glow(in_channel=3, L=5, K=16, compute="bf16")
    flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 1, DEEP))) for k, v in flow.state_dict().items()})
    flow.to(dev)
    G.set_weight_term_cache(False)
    xh = (torch.rand(B, 128, 128, 3, device=dev) - 0.5)
    def fwd():
        flow.transform_nhwc(xh, torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev))
    for mode in ("auto", "always", "never"):
        for fb in (True, False):
            K.set_mlp_rows(mode); G.FUSED_LEVEL_BOUNDARY = fb
            for _ in range(3): fwd()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): fwd()
            for _ in range(20): g.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): g.replay()
            torch.cuda.synchronize()
            print(f"mlp rows {mode:7s} fused boundary {fb}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
