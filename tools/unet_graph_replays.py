import os, sys, time
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
    for _ in range(3): u.forward_nhwc(x, t)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        u.forward_nhwc(x, t)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            y = u.forward_nhwc(x, t)
    torch.cuda.synchronize()
    for _ in range(400): g.replay()
    torch.cuda.synchronize()
