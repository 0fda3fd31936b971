"""GaussianDiffusion options off the reference drivers' default path - objectives pred_x0 / pred_v, self-conditioning, auto_normalize
(gaussian_diffusion.py:647-663, :674-694, :726-728, :744-747, :832-855, :148-157) - against golden vectors captured by running the
reference (tests/golden/diffusion_variants.npz, tools/make_goldens.py variants) with all randomness injected.  1e-4 relative."""
import pytest
import torch

from tests.util import load_golden, meta_of, synth_sd, T, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


@pytest.fixture(scope="module")
def vg():
    return load_golden("diffusion_variants.npz")


def make_unet(vg, self_condition=False):
    from diffusion_prior import Unet
    u = Unet(dim=64, dim_mults=(1, 2), channels=2, resnet_block_groups=8, self_condition=self_condition)
    u.load_state_dict(synth_sd(meta_of(vg)["unet_sc" if self_condition else "unet"], 24 if self_condition else 21), strict=True)
    return u.to(DEV)


def make_diffusion(unet, timesteps=1000, **kw):
    from diffusion_prior import GaussianDiffusion
    kw.setdefault("auto_normalize", False)
    return GaussianDiffusion(unet, image_size=16, timesteps=timesteps, beta_schedule="cosine", loss_type="l2", **kw).to(DEV)


def check_loops(vg, tag, unet, ddim_tol=TOL, **kw):
    d8 = make_diffusion(unet, timesteps=8, **kw)
    out = d8.p_sample_loop((2, 2, 16, 16), x_T=T(vg[f"{tag}_loop8_xT"]), noises=list(T(vg[f"{tag}_loop8_noises"])))
    assert_close(out, vg[f"{tag}_loop8_out"], TOL, f"{tag}: p_sample_loop T=8")
    dd = make_diffusion(unet, timesteps=8, sampling_timesteps=4, ddim_sampling_eta=1.0, **kw)
    out = dd.ddim_sample((2, 2, 16, 16), x_T=T(vg[f"{tag}_ddim_xT"]), noises=list(T(vg[f"{tag}_ddim_noises"])))
    assert_close(out, vg[f"{tag}_ddim_out"], ddim_tol, f"{tag}: ddim T=8 S=4")
    for d in (d8, dd):                                             # device-RNG entry points (captured steps where they apply)
        s = d.sample(batch_size=3)
        assert s.shape == (3, 2, 16, 16) and torch.isfinite(s).all()
    return d8, dd


@pytest.mark.parametrize("obj", ["pred_x0", "pred_v"])
@torch.no_grad()
def test_objectives_golden(vg, obj):
    u = make_unet(vg)
    d = make_diffusion(u, objective=obj)
    x0, eps, tt = T(vg["x0"]).to(DEV), T(vg["eps"]).to(DEV), T(vg["t"]).to(DEV)
    assert_close(d.p_losses(x0, tt, noise=eps), vg[f"{obj}_p_losses"], TOL, "p_losses")
    for clip in (False, True):
        mp = d.model_predictions(x0, tt, clip_x_start=clip)
        assert_close(mp.pred_x_start, vg[f"{obj}_mp_x0_clip{int(clip)}"], TOL, "model_predictions x_start")
        assert_close(mp.pred_noise, vg[f"{obj}_mp_noise_clip{int(clip)}"], TOL, "model_predictions pred_noise")
    for step in (0, 500):
        img, xs = d.p_sample(x0, step, noise=T(vg[f"{obj}_psample{step}_noise"]).to(DEV))
        assert_close(img, vg[f"{obj}_psample{step}_img"], TOL, f"p_sample t={step}")
        assert_close(xs, vg[f"{obj}_psample{step}_x0"], TOL, f"p_sample x_start t={step}")
    assert_close(d.predict_v(x0, tt, eps), vg[f"{obj}_predict_v"], 1e-6, "predict_v")
    assert_close(d.predict_noise_from_start(x0, tt, eps), vg[f"{obj}_noise_from_start"], 1e-6, "predict_noise_from_start")
    d8, dd = check_loops(vg, obj, u, objective=obj)
    # the captured DDIM step serves every objective: same table-driven kernel as the eager loop
    assert any(k[0] == "ddim" for k in dd._graphs) and any(k[0] == "step" for k in d8._graphs)


@torch.no_grad()
def test_self_conditioning_golden(vg, monkeypatch):
    import diffusion_prior.gaussian_diffusion as GD
    u = make_unet(vg, self_condition=True)
    assert u.init_conv.weight.shape[1] == 4
    x0, eps, tt, sc = T(vg["x0"]).to(DEV), T(vg["eps"]).to(DEV), T(vg["t"]).to(DEV), T(vg["sc"]).to(DEV)
    assert_close(u(x0, tt, sc), vg["sc_unet_given"], TOL, "U-Net with x_self_cond")
    assert_close(u(x0, tt), vg["sc_unet_none"], TOL, "U-Net with the zero default")
    d = make_diffusion(u)
    assert d.self_condition
    for name, draw in (("taken", 0.1), ("skipped", 0.9)):
        monkeypatch.setattr(GD, "_random", lambda draw=draw: draw)       # the reference draws python's random() (:838)
        assert_close(d.p_losses(x0, tt, noise=eps), vg[f"sc_p_losses_{name}"], TOL, f"p_losses, self-conditioning {name}")
    img, xs = d.p_sample(x0, 500, sc, noise=T(vg["sc_psample500_noise"]).to(DEV))
    assert_close(img, vg["sc_psample500_img"], TOL, "p_sample with x_self_cond")
    assert_close(xs, vg["sc_psample500_x0"], TOL, "p_sample x_start")
    # DDIM with self-conditioning feeds the x_start prediction of the first step (t = 7 of the 8-step cosine schedule: sqrt(1/ac) = 163, so
    # fp32 noise of the U-Net output arrives in x_start 163 times larger) back through the U-Net: a chain that amplifies rounding
    # differences of ANY two fp32 implementations; it is held to 1e-3 (observed 3e-4), every single step above to 1e-4
    check_loops(vg, "sc", u, ddim_tol=1e-3)


def test_self_conditioning_training_path(vg, monkeypatch):
    """p_losses under autograd with self-conditioning: the conditioning prediction carries no gradient, the loss matches the
    inference path, every U-Net parameter (incl. the 4-channel stem) gets a finite gradient."""
    import diffusion_prior.gaussian_diffusion as GD
    u = make_unet(vg, self_condition=True)
    d = make_diffusion(u, objective="pred_v")
    x0, eps, tt = T(vg["x0"]).to(DEV), T(vg["eps"]).to(DEV), T(vg["t"]).to(DEV)
    monkeypatch.setattr(GD, "_random", lambda: 0.1)
    with torch.no_grad():
        ref = d.p_losses(x0, tt, noise=eps)
    loss = d.p_losses(x0, tt, noise=eps)
    loss.backward()
    assert_close(loss.detach(), ref, 2e-5, "training-path loss")
    grads = [p.grad for p in u.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    assert float(u.init_conv.weight.grad[:, :2].abs().max()) > 0 and float(u.init_conv.weight.grad[:, 2:].abs().max()) > 0


@torch.no_grad()
def test_auto_normalize_golden(vg):
    u = make_unet(vg)
    d = make_diffusion(u, timesteps=8, auto_normalize=True)
    x01 = T(vg["an_x01"]).to(DEV)
    loss = d.p_losses(d.normalize(x01), T(vg["an_fwd_t"]).to(DEV), noise=T(vg["an_fwd_noise"]).to(DEV))   # forward() with its draws injected
    assert_close(loss, vg["an_fwd_loss"], TOL, "forward() on [0,1] images")
    assert_close(d.normalize(x01), x01 * 2 - 1, 1e-7, "normalize")
    assert_close(d.unnormalize(x01 * 2 - 1), x01, 1e-6, "unnormalize")
    torch.manual_seed(0)
    s = d.sample(batch_size=2)
    assert s.shape == (2, 2, 16, 16) and torch.isfinite(s).all()
    assert float(s.min()) >= -1e-6 and float(s.max()) <= 1 + 1e-6      # the last step clamps x_start to [-1,1] before unnormalize


# ----------------------------------------------------------------------------- random / learned sinusoidal time embeddings
@pytest.mark.parametrize("tag", ["learned", "random"])
def test_unet_with_learned_or_random_time_embedding(tag):
    """Unet(learned_sinusoidal_cond=True) / Unet(random_fourier_features=True) (gaussian_diffusion.py:240-255, :398-405): embedding and
    U-Net output against the reference (tests/golden/unet_embeddings.npz); the learned frequencies' gradient (and two other parameters'
    through the same graph) against the reference's autograd; GaussianDiffusion rejects such a U-Net as the reference does (:559)."""
    from diffusion_prior import Unet, GaussianDiffusion
    eg = load_golden("unet_embeddings.npz")
    kw = dict(learned_sinusoidal_cond=True) if tag == "learned" else dict(random_fourier_features=True)
    u = Unet(dim=64, dim_mults=(1, 2), channels=2, resnet_block_groups=8, learned_sinusoidal_dim=16, **kw)
    u.load_state_dict(synth_sd(meta_of(eg)[tag], 26), strict=True)
    u = u.to(DEV)
    assert u.time_mlp[0].weights.requires_grad == bool(int(eg[f"{tag}_requires_grad"]))
    x0, eps, t = T(eg["x0"]).to(DEV), T(eg["eps"]).to(DEV), T(eg["t"]).to(DEV)
    with torch.no_grad():
        assert_close(u.time_mlp[0](t), eg[f"{tag}_emb"], 1e-5, "embedding")
        assert_close(u(x0, t), eg[f"{tag}_out"], TOL, "U-Net output (inference path)")
    out = u(x0, t)                                                  # grad mode: the differentiable path
    assert_close(out, eg[f"{tag}_out"], TOL, "U-Net output (training path)")
    (out * eps).sum().backward()
    if tag == "learned":
        assert_close(u.time_mlp[0].weights.grad, eg["learned_grad_weights"], 2e-3, "d loss / d frequencies")
        assert_close(u.time_mlp[1].weight.grad, eg["learned_grad_lin1"], 2e-3, "d loss / d time_mlp.1.weight")
        assert_close(u.init_conv.weight.grad, eg["learned_grad_init_conv"], 2e-3, "d loss / d init_conv.weight")
    else:
        assert u.time_mlp[0].weights.grad is None
    with pytest.raises(AssertionError):
        GaussianDiffusion(u, image_size=16, timesteps=8)


def test_auto_normalize_keeps_the_gradient_of_a_latent_that_requires_grad():
    """GaussianDiffusion(auto_normalize=True).forward on an input that carries a gradient (latents of an unfrozen flow,
    diffusion_prior/trainer.py:139-141): the [0,1] -> [-1,1] map must stay inside autograd (d loss / d img = 2 * d loss / d x_start)."""
    from diffusion_prior import Unet, GaussianDiffusion
    u = Unet(dim=32, dim_mults=(1, 2), channels=2, resnet_block_groups=8, compute="f32").to(DEV)
    d = GaussianDiffusion(u, image_size=8, timesteps=20, beta_schedule="cosine", loss_type="l2", auto_normalize=True).to(DEV)
    g = torch.Generator().manual_seed(3)
    for p in u.parameters():
        p.data.add_(0.05 * torch.randn(p.shape, generator=g).to(DEV))
    img = torch.rand(2, 2, 8, 8, generator=g).to(DEV).requires_grad_(True)
    t = torch.tensor([3, 11], device=DEV)
    eps = torch.randn(2, 2, 8, 8, generator=g).to(DEV)
    loss = d.p_losses(d.normalize(img), t, noise=eps)
    loss.backward()
    assert img.grad is not None and float(img.grad.abs().max()) > 0
    x = (img.detach() * 2 - 1).requires_grad_(True)
    d.p_losses(x, t, noise=eps).backward()
    assert_close(img.grad, 2 * x.grad, 1e-5, "gradient through normalize")
    with torch.no_grad():
        assert_close(d.normalize(img.detach()), img.detach() * 2 - 1, 1e-7, "HIP normalize (no grad)")
