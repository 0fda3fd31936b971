#!/usr/bin/env python3
"""Training-step timing (BASELINE configs[2] shape on one GPU): Glow L=3 K=16 on 32x32x3, B images per step, the loop
body of normalizing_flow/trainer.py:152-167 (transform -> bits/dim -> backward -> clips -> Adam step) on the HIP
training path (exact fp32).  usage: python tools/bench_train.py [B] [steps] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "normalizing-flow-with-diffusion-prior-model_amd")]
import numpy as np, torch
import nfdp_hip as K
from tests.golden.synth import synth_tensor, DEEP
from normalizing_flow import Glow, GaussianPrior, calculate_loss, initialize_with_zeros, init_optimizer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Kf = int(sys.argv[3]) if len(sys.argv) > 3 else 16
COMPUTE = sys.argv[4] if len(sys.argv) > 4 else "f32"
dev = torch.device("cuda", 0)
flow = Glow(in_channel=3, L=3, K=Kf, learn_prior_mean_logs=True, compute=COMPUTE)
prior = GaussianPrior(48, True)
flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 1, DEEP))) for k, v in flow.state_dict().items()})
prior.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(k, v.shape, 2, DEEP))) for k, v in prior.state_dict().items()})
flow.to(dev); prior.to(dev)
opt = init_optimizer("adam", [{"params": list(flow.parameters())}, {"params": list(prior.parameters())}], lr=1e-4)
g = torch.Generator().manual_seed(0)
x = ((torch.randint(0, 256, (B, 3, 32, 32), generator=g) // 8).float() / 32.0 - 0.5 + torch.rand(B, 3, 32, 32, generator=g) / 32.0).to(dev)


def step():
    ll, logp = initialize_with_zeros(2, B, dev)
    latents, ll, logp = flow.transform(x, ll, logp)
    logp = logp + prior.compute_log_prob(latents[-1])
    loss = calculate_loss(ll + logp, 32.0, 32 * 32 * 3.0)
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_value_(flow.parameters(), 1)
    torch.nn.utils.clip_grad_norm_(flow.parameters(), 1)
    opt.step()
    return loss


for i in range(2):
    l = step()
torch.cuda.synchronize()
print(f"warm-up done, loss {float(l.detach()):.4f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
t0 = time.perf_counter()
for i in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"B={B} K={Kf} {COMPUTE}: {dt * 1e3:.1f} ms/step, {B / dt:.0f} images/s, loss {float(l.detach()):.4f}", flush=True)
K.start_trace()
step()
recs = K.stop_trace()
agg = {}
for r in recs:
    key = r["label"].split(" hw")[0]
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
tot = sum(a[1] for a in agg.values())
print(f"traced step: {len(recs)} launches through the C ABI, {tot:.1f} ms of kernel time")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    tf = f", {a[2] / a[1] / 1e9:.1f} TFLOP/s" if a[2] else ""
    print(f"  {a[1]:8.2f} ms  x{a[0]:4d}  {key}{tf}")
