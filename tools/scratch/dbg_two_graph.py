import sys; sys.path[:0]=[".","normalizing-flow-with-diffusion-prior-model_amd"]
import torch
from tests.util import T
from tests.golden.synth import synth_tensor, synth_input, DEEP
from normalizing_flow import Glow, GaussianPrior, calculate_loss
from nfdp_hip.optim import FusedAdam
from nfdp_hip.graphs import capture_step
DEV="cuda"; B=8
x = (T(synth_input("two_graph_x", (B, 3, 32, 32), 4)) * 0.4).to(DEV)
def make(lr):
    flow, prior = Glow(in_channel=3, L=3, K=2, compute="bf16").to(DEV), GaussianPrior(48).to(DEV)
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 6, DEEP)) for k, v in flow.state_dict().items()})
    prior.load_state_dict({k: T(synth_tensor(k, v.shape, 6, DEEP)) for k, v in prior.state_dict().items()})
    opt = FusedAdam(clipped=flow.parameters(), unclipped=prior.parameters(), lr=lr, clip_value=1, max_norm=1, direct_grads=True)
    def fwd_bwd():
        ldj = torch.zeros(B, dtype=torch.float64, device=DEV); logp = torch.zeros(B, dtype=torch.float64, device=DEV)
        zs, ldj, logp = flow.transform(x, ldj, logp)
        loss = calculate_loss(ldj + logp + prior.compute_log_prob(zs[-1]), 32.0, 3072.0)
        opt.zero_grad(); loss.backward(); return loss.detach()
    def step():
        l = fwd_bwd(); opt.step(); return l
    return opt, fwd_bwd, step
for lr in (1e-3, 1e-4):
    oe,_,eager = make(lr); print("lr",lr,"eager:", [round(float(eager()),5) for _ in range(6)])
    o1,_,s1 = make(lr); g1 = capture_step(s1, warmup=2); print("   one graph (2 warm-up):", [round(float(g1()),5) for _ in range(4)])
    oc,fb,cs = make(lr)
    g_fb = capture_step(fb, warmup=1, mutates_params=False, warmup_fn=cs); g_opt = capture_step(oc.step, warmup=1, warmup_fn=cs)
    out=[]
    for _ in range(4):
        l=g_fb(); g_opt(); out.append(round(float(l),5))
    print("   two graphs (2 warm-up):", out)
