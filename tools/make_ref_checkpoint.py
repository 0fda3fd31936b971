#!/usr/bin/env python3
"""tests/golden/ref_ckpt/: a checkpoint WRITTEN BY THE REFERENCE - its own `save_model` (normalizing_flow/prior.py:102-115) on its own
Glow / GaussianPrior / torch Adam after one training step (so the optimizer state is populated), plus `architecture.pkl` as
run_baseline_experiment.py:65-68 pickles it - and `ref_ckpt.npz`: the input batch and the reference's outputs for the checkpointed
weights (latents, log-det, log-prob).  The files are data produced by running the reference; no reference source is stored.

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python tools/make_ref_checkpoint.py
Glow hard-wires the 512-wide coupling networks (glow.py:87, :169), so the smallest model (1 channel, L = 2, K = 1) is 0.6 M
parameters: 2.3 MB of weights + 4.6 MB of Adam moments."""
import json
import logging
import os
import pickle
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

from ref_import import import_reference
from tests.golden.synth import images_5bit

R = import_reference()
OUT = os.path.join(ROOT, "tests", "golden", "ref_ckpt")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)
torch.manual_seed(20261005)

C, L, K_, B = 1, 2, 1, 4
arch = {"L": L, "K": K_, "learn_prior_mean_logs": True}
flow = R["glow"].Glow(in_channel=C, L=L, K=K_, learn_prior_mean_logs=True)
prior = R["prior"].GaussianPrior(2 ** (L + 1) * C, True)
U = R["nf_utils"]
optim = U.init_optimizer("adam", [{"params": list(flow.parameters()) + list(prior.parameters())}], lr=1e-3)
x_np, _, _ = images_5bit("ref_ckpt", (B, C, 16, 16), 77)
x = torch.from_numpy(x_np)
n_pixel = 16 * 16 * 3.0                                     # the reference's n_pixel quirk (trainer.py:134)
for it in range(2):                                         # trainer.py:152-167: the first step also runs the data-dependent init
    ll, logp = U.initialize_with_zeros(2, B, torch.device("cpu"))
    lat, ll, logp = flow.transform(x, ll, logp)
    logp = logp + prior.compute_log_prob(lat[-1])
    loss = U.calculate_loss(ll + logp, 32.0, n_pixel)
    optim.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_value_(flow.parameters(), 1)
    torch.nn.utils.clip_grad_norm_(flow.parameters(), 1)
    optim.step()
R["prior"].save_model(logging.getLogger("ref"), flow, prior, optim, 0, 2, OUT)           # -> model_gaussian_000.pt
with open(os.path.join(OUT, "architecture.pkl"), "wb") as f:                               # run_baseline_experiment.py:65-68
    pickle.dump(arch, f)
with torch.no_grad():
    ll, logp = U.initialize_with_zeros(2, B, torch.device("cpu"))
    lat, ll, logp = flow.transform(x, ll, logp)
    logp = logp + prior.compute_log_prob(lat[-1])
    bpd = U.calculate_loss(ll + logp, 32.0, n_pixel)
    x_rec = flow.invert(lat)
np.savez_compressed(os.path.join(OUT, "ref_ckpt.npz"), x=x_np, ldj=ll.numpy(), logp=logp.numpy(), bpd=np.array(float(bpd)),
                    x_rec=x_rec.numpy(), **{f"z{i}": z.numpy() for i, z in enumerate(lat)},
                    meta=json.dumps({"C": C, "L": L, "K": K_, "B": B, "size": 16, "epoch": 0, "current_iter": 2,
                                     "flow_keys": list(flow.state_dict().keys()), "prior_keys": list(prior.state_dict().keys())}))
for f in sorted(os.listdir(OUT)):
    print(f, os.path.getsize(os.path.join(OUT, f)))
