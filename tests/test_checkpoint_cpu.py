"""Checkpoint I/O compatibility: files are written/read with the reference's names and keys
(normalizing_flow/prior.py:102-115, run_baseline_experiment.py:65-68, :112-114, normalizing_flow/__init__.py:55-57), and
the diffusion checkpoint actually contains the U-Net weights (the reference's `diffusion_prior.state_dict()` is empty
because its per-part models sit in a plain list, model.py:100-102)."""
import logging
import os

import numpy as np
import torch

from tests.golden.synth import synth_tensor


def fill(module, seed):
    """Distinct deterministic values in every parameter; schedule buffers (no synthesis rule) keep their values."""
    sd = {}
    for k, v in module.state_dict().items():
        try:
            sd[k] = torch.from_numpy(np.array(synth_tensor(k, v.shape, seed)))
        except KeyError:
            sd[k] = v.detach().cpu().clone()
    module.load_state_dict(sd, strict=True)
    return sd


def test_gaussian_checkpoint_roundtrip(tmp_path):
    from normalizing_flow import (Glow, GaussianPrior, NFBackbone, save_model, load_model, save_architecture,
                                  load_architecture, init_optimizer)
    flow, prior = Glow(in_channel=1, L=2, K=1, learn_prior_mean_logs=True), GaussianPrior(4, True)
    sd_f, sd_p = fill(flow, 3), fill(prior, 4)
    optim = init_optimizer("adam", [{"params": list(flow.parameters()) + list(prior.parameters())}], lr=1e-3)
    save_model(logging.getLogger("t"), flow, prior, optim, 7, 123, str(tmp_path))
    path = os.path.join(tmp_path, "model_gaussian_007.pt")
    assert os.path.exists(path)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"flow", "prior_dist", "optimizer", "current_iter"} and ck["current_iter"] == 123
    flow2, prior2 = Glow(in_channel=1, L=2, K=1, learn_prior_mean_logs=True), GaussianPrior(4, True)
    assert load_model(flow2, prior2, str(tmp_path), 7) == 123
    for k, v in sd_f.items():
        assert torch.equal(flow2.state_dict()[k].cpu(), v), k
    for k, v in sd_p.items():
        assert torch.equal(prior2.state_dict()[k].cpu(), v), k
    # the reference's own consumer of this file: NFBackbone(model_dir=...)
    nf = NFBackbone(model_dir=path, in_channel=1, L=2, K=1, learn_prior_mean_logs=True, freeze_flow=True)
    assert all(torch.equal(nf.model.state_dict()[k].cpu(), v) for k, v in sd_f.items())
    assert not any(p.requires_grad for p in nf.parameters())
    save_architecture(str(tmp_path), 2, 1, True)
    assert load_architecture(str(tmp_path)) == {"L": 2, "K": 1, "learn_prior_mean_logs": True}


def test_diffusion_checkpoint_contains_unets(tmp_path):
    from normalizing_flow import NFBackbone, save_model, load_model, init_optimizer
    from diffusion_prior import DiffusionPrior, IdentityFormater

    def build():
        nf = NFBackbone(model_dir="", in_channel=1, L=2, K=1, learn_prior_mean_logs=True, freeze_flow=True)
        fmt = IdentityFormater(2, 1, 16)
        shapes = fmt.get_input_shapes()
        dp = DiffusionPrior(latent_formater=fmt, unet_kwargs={"dim": 8, "dim_mults": (1, 2), "resnet_block_groups": 4},
                            diffusion_kwargs={"timesteps": 4, "beta_schedule": "cosine", "loss_type": "l2"},
                            image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes])
        return nf, dp

    nf, dp = build()
    sd_n, sd_d = fill(nf, 5), fill(dp, 6)
    assert any(k.startswith("priors.0.") and "model" in k for k in sd_d), "U-Net weights missing from the state dict"
    optim = init_optimizer("adamw", [{"params": list(dp.parameters())}], lr=1e-4)
    save_model(logging.getLogger("t"), nf, dp, optim, 2, 9, str(tmp_path))
    assert os.path.exists(os.path.join(tmp_path, "model_diffusion_002.pt"))
    nf2, dp2 = build()
    assert load_model(nf2, dp2, str(tmp_path), 2, optim=optim) == 9
    assert all(torch.equal(dp2.state_dict()[k].cpu(), v) for k, v in sd_d.items())
    assert all(torch.equal(nf2.state_dict()[k].cpu(), v) for k, v in sd_n.items())
