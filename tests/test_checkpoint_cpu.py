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


REF_CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ckpt")


def test_checkpoint_written_by_the_reference_loads():
    """tests/golden/ref_ckpt/model_gaussian_000.pt + architecture.pkl were written by the REFERENCE's own save_model
    (normalizing_flow/prior.py:102-115) and pickle lines (run_baseline_experiment.py:65-68) after two of its training steps
    (tools/make_ref_checkpoint.py): the file format - dict keys, state-dict keys and shapes, a populated torch-Adam state, the pickle -
    is read by this package's load_architecture / load_model / NFBackbone and the optimizer state by init_optimizer's Adam."""
    import json
    from normalizing_flow import Glow, GaussianPrior, NFBackbone, load_model, load_architecture, init_optimizer
    arch = load_architecture(REF_CKPT)
    assert arch == {"L": 2, "K": 1, "learn_prior_mean_logs": True}
    meta = json.loads(str(np.load(os.path.join(REF_CKPT, "ref_ckpt.npz"))["meta"]))
    ck = torch.load(os.path.join(REF_CKPT, "model_gaussian_000.pt"), map_location="cpu")
    assert set(ck) == {"flow", "prior_dist", "optimizer", "current_iter"} and ck["current_iter"] == meta["current_iter"]
    flow = Glow(in_channel=meta["C"], L=arch["L"], K=arch["K"], learn_prior_mean_logs=arch["learn_prior_mean_logs"])
    prior = GaussianPrior(2 ** (arch["L"] + 1) * meta["C"], arch["learn_prior_mean_logs"])
    assert list(flow.state_dict().keys()) == meta["flow_keys"] == list(ck["flow"].keys())       # same keys, same ORDER
    assert list(prior.state_dict().keys()) == meta["prior_keys"] == list(ck["prior_dist"].keys())
    optim = init_optimizer("adam", [{"params": list(flow.parameters()) + list(prior.parameters())}], lr=1e-3)
    assert load_model(flow, prior, REF_CKPT, 0, optim=optim) == 2
    for k, v in ck["flow"].items():
        assert torch.equal(flow.state_dict()[k].cpu(), v), k
    for k, v in ck["prior_dist"].items():
        assert torch.equal(prior.state_dict()[k].cpu(), v), k
    # the reference's Adam state arrived: one (step, exp_avg, exp_avg_sq) per parameter, in the parameter order both sides share
    params = list(flow.parameters()) + list(prior.parameters())
    st = optim.state_dict()["state"]
    assert len(st) == len(params) == len(ck["optimizer"]["state"])
    for i, p in enumerate(params):
        assert st[i]["exp_avg"].shape == p.shape and float(st[i]["step"]) == 2.0
        assert torch.equal(st[i]["exp_avg_sq"].cpu(), ck["optimizer"]["state"][i]["exp_avg_sq"])
    assert all(bool(f.actnorm.is_initialized) for f in flow.final_flows)       # the data-dependent init flags travelled too
    nf = NFBackbone(model_dir=os.path.join(REF_CKPT, "model_gaussian_000.pt"), in_channel=meta["C"], freeze_flow=True, **arch)
    assert all(torch.equal(nf.model.state_dict()[k].cpu(), v) for k, v in ck["flow"].items())
