import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import StepFlow
dev = torch.device("cuda", 0)
with torch.no_grad():
    for c, hw in ((96, 8), (192, 4)):
        m = StepFlow(c, 512, compute="bf16")
        m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor("flows.0." + k, v.shape, 1, DEEP))) for k, v in m.state_dict().items()})
        m = m.to(dev)
        x = (torch.rand(5, hw, hw, c, device=dev) - 0.5)
        res = {}
        for fused in (True, False):
            m.affcoupling.fused = fused
            y = x.clone(); ldj = torch.zeros(5, dtype=torch.float64, device=dev)
            y = m.transform_nhwc_(y, ldj, None, True)
            xr = m.invert_nhwc_(y.clone())
            res[fused] = (y.clone(), ldj.clone(), xr.clone())
        yf, lf, xf = res[True]; yu, lu, xu = res[False]
        print(f"C={c}: fwd fused vs unfused {float((yf - yu).abs().max() / yu.abs().max()):.2e}; ldj {float((lf - lu).abs().max() / lu.abs().max()):.2e}; "
              f"round trip fused {float((xf - x).abs().max()):.2e}, unfused {float((xu - x).abs().max()):.2e}; "
              f"inverse of the SAME y: fused vs unfused ", end="")
        m.affcoupling.fused = True; a = m.invert_nhwc_(yu.clone())
        m.affcoupling.fused = False; b = m.invert_nhwc_(yu.clone())
        print(f"{float((a - b).abs().max() / b.abs().max()):.2e}")
    # the level path (several flows in one launch) of the C = 96 level against flow-by-flow unfused
    from normalizing_flow.glow import _run_flows_fwd, _run_flows_inv
    for nfl in (1, 2, 3):
        flows = []
        for f in range(nfl):
            m = StepFlow(96, 512, compute="bf16")
            m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(f"flows.{f}." + k, v.shape, 1, DEEP))) for k, v in m.state_dict().items()})
            flows.append(m.to(dev))
        x = (torch.rand(5, 8, 8, 96, device=dev) - 0.5)
        out = {}
        for fused in (True, False):
            for m in flows:
                m.affcoupling.fused = fused
            y = _run_flows_fwd(flows, x.clone(), None)
            xr = _run_flows_inv(flows, y.clone())
            out[fused] = (y, xr)
        for m in flows:
            m.affcoupling.fused = True
        xi = _run_flows_inv(flows, out[False][0].clone())
        print(f"level path, {nfl} flows: fwd fused vs unfused {float((out[True][0] - out[False][0]).abs().max() / out[False][0].abs().max()):.2e}; "
              f"round trip fused {float((out[True][1] - x).abs().max()):.2e} unfused {float((out[False][1] - x).abs().max()):.2e}; "
              f"fused inverse of the unfused y vs x {float((xi - x).abs().max()):.2e}")
