#!/usr/bin/env python3
"""Shape sweep (diagnostic): Glow forward+NLL / inverse / one training step for model geometries beyond the two benchmarked
ones, each checked against the oracle (forward) and for finiteness (inverse round trip, gradients)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from oracle import nf_oracle as O
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import Glow, GaussianPrior, calculate_loss
dev = "cuda"
for (c, L, kf, size, b) in [(1, 5, 1, 32, 3), (3, 4, 1, 64, 2), (1, 4, 2, 64, 2), (3, 2, 2, 16, 5), (3, 5, 1, 64, 2), (1, 3, 2, 8, 4)]:
    for compute in ("f32", "bf16"):
        flow = Glow(in_channel=c, L=L, K=kf, compute=compute).to(dev)
        prior = GaussianPrior(2 ** (L + 1) * c).to(dev)
        flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 3, DEEP))) for k, v in flow.state_dict().items()})
        prior.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 3, DEEP))) for k, v in prior.state_dict().items()})
        g = torch.Generator().manual_seed(c * 100 + L)
        x = (torch.rand(b, c, size, size, generator=g) - 0.5)
        try:
            with torch.no_grad():
                ldj = torch.zeros(b, dtype=torch.float64, device=dev); logp = torch.zeros(b, dtype=torch.float64, device=dev)
                zs, _, _ = flow.transform(x.to(dev), ldj, logp)
                logp = logp + prior.compute_log_prob(zs[-1])
                sd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
                psd = {k: v.detach().cpu() for k, v in prior.state_dict().items()}
                zo, ldj_o, logp_o, _ = O.fwd_nll(x, sd, psd, L, kf)
                ez = max(float((a.cpu() - bb).abs().max() / bb.abs().max()) for a, bb in zip(zs, zo))
                el = float((ldj.cpu() - ldj_o).abs().max() / ldj_o.abs().max())
                ep = float((logp.cpu() - logp_o).abs().max() / logp_o.abs().max())
                rt = float((flow.invert(zs).cpu() - x).abs().max() / x.abs().max())
            ldj = torch.zeros(b, dtype=torch.float64, device=dev); logp = torch.zeros(b, dtype=torch.float64, device=dev)
            zs, ldj, logp = flow.transform(x.to(dev), ldj, logp)
            loss = calculate_loss(ldj + logp + prior.compute_log_prob(zs[-1]), 32.0, size * size * 3.0)
            loss.backward()
            gfin = all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in flow.parameters())
            print(f"C={c} L={L} K={kf} {size}x{size} B={b} {compute}: z {ez:.1e} ldj {el:.1e} logp {ep:.1e} round-trip {rt:.1e} grads finite {gfin}", flush=True)
        except Exception as e:       # noqa
            print(f"C={c} L={L} K={kf} {size}x{size} B={b} {compute}: FAILED {type(e).__name__}: {str(e)[:160]}", flush=True)
