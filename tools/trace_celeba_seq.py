import os, sys, collections
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import Glow
B = 32
dev = torch.device("cuda", 0)
with torch.no_grad():
    flow = Glow(in_channel=3, L=5, K=16, compute="bf16")
    flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 1, DEEP))) for k, v in flow.state_dict().items()})
    flow.to(dev)
    xh = (torch.rand(B, 128, 128, 3, device=dev) - 0.5)
    for _ in range(2):
        flow.transform_nhwc(xh, torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev))
    K.start_trace()
    flow.transform_nhwc(xh, torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev))
    tr = K.stop_trace()
for i, e in enumerate(tr):
    print(f"{i:4d} {e['ms']*1e3:8.1f} us  {e['label']}")
