import os, sys
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import torch
import nfdp_hip as K
from diffusion_prior import Unet
dev = torch.device("cuda", 0)
with torch.no_grad():
    u = Unet(dim=64, dim_mults=(1, 2), channels=2, resnet_block_groups=8).to(dev)
    x = torch.randn(64, 16, 16, 2, device=dev)
    t = torch.randint(0, 1000, (64,), device=dev)
    for _ in range(2): u.forward_nhwc(x, t)
    K.start_trace()
    u.forward_nhwc(x, t)
    ev = K.stop_trace()
    for i, e in enumerate(ev):
        print(f"{i:3d} {e['ms']*1e3:7.1f} us  {e['label']}")
