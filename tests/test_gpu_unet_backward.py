"""Backward kernels of the U-Net glue ops and the differentiable U-Net / p_losses path (training the
diffusion prior) against torch autograd: per op on the CPU formulation of the oracle, and for the whole U-Net + loss against
autograd through oracle.diffusion_oracle on the same weights.  fp32, 1e-4 relative to each gradient tensor's largest entry."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import nfdp_hip as K
from oracle import diffusion_oracle as D
from tests.util import T, assert_close, relerr, load_golden, meta_of, synth_sd

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(a):
    return a.detach().permute(0, 2, 3, 1).contiguous().to(DEV)


@pytest.mark.parametrize("b,c,hw,groups,with_ss", [(2, 64, 8, 8, True), (3, 128, 4, 8, False), (2, 192, 4, 8, True), (1, 64, 16, 16, True)])
def test_groupnorm_silu_bwd(b, c, hw, groups, with_ss):
    x = rnd(b, c, hw, hw, seed=1).requires_grad_()
    gamma = (1 + 0.2 * rnd(c, seed=2)).requires_grad_()
    beta = (0.1 * rnd(c, seed=3)).requires_grad_()
    ss = (0.3 * rnd(b, 2 * c, seed=4)).requires_grad_() if with_ss else None
    y = F.group_norm(x, groups, gamma, beta, eps=1e-5)
    if with_ss:
        sc, sh = ss[:, :c, None, None], ss[:, c:, None, None]
        y = y * (sc + 1) + sh
    y = F.silu(y)
    dy = rnd(b, c, hw, hw, seed=5)
    y.backward(dy)
    wide = torch.zeros(b, 2 * c + 7, device=DEV)                       # scale_shift as a column slice of a wider buffer
    if with_ss:
        wide[:, 3:3 + 2 * c] = ss.detach().to(DEV)
    dx, dg, db, dss = K.groupnorm_silu_bwd(nhwc(x), nhwc(dy), groups, gamma.detach().to(DEV), beta.detach().to(DEV),
                                           wide[:, 3:3 + 2 * c] if with_ss else None, 1e-5)
    assert_close(dx.permute(0, 3, 1, 2), x.grad, TOL, "dx")
    assert_close(dg, gamma.grad, TOL, "dgamma")
    assert_close(db, beta.grad, TOL, "dbeta")
    if with_ss:
        assert_close(dss, ss.grad, TOL, "dscale_shift")
    else:
        assert dss is None


@pytest.mark.parametrize("c", [64, 128, 20])
def test_chan_layernorm_bwd(c):
    x = rnd(2, c, 4, 4, seed=6).requires_grad_()
    g = (1 + 0.3 * rnd(c, seed=7)).requires_grad_()
    y = D.chan_layernorm(x, g.view(1, c, 1, 1))
    dy = rnd(2, c, 4, 4, seed=8)
    y.backward(dy)
    dx, dg = K.chan_layernorm_bwd(nhwc(x), nhwc(dy), g.detach().to(DEV))
    assert_close(dx.permute(0, 3, 1, 2), x.grad, TOL, "dx")
    assert_close(dg, g.grad, TOL, "dgain")


@pytest.mark.parametrize("hw", [4, 8, 16])
def test_attention_cores_bwd(hw):
    b, heads, d = 2, 4, 32
    n = hw * hw
    qkv = rnd(b, 3 * heads * d, hw, hw, seed=9, scale=0.8).requires_grad_()
    dout = rnd(b, heads * d, hw, hw, seed=10)
    q, k, v = (t.reshape(b, heads, d, n) for t in qkv.chunk(3, dim=1))
    # linear attention core (oracle: gaussian_diffusion.py:320-332)
    out = torch.einsum("bhde,bhdn->bhen", torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v / n), q.softmax(dim=-2) * d ** -0.5)
    out.reshape(b, heads * d, hw, hw).backward(dout)
    got = K.linear_attention_bwd(nhwc(qkv), nhwc(dout), heads, d, d ** -0.5)
    assert_close(got.permute(0, 3, 1, 2), qkv.grad, TOL, "linear attention dqkv")
    # full attention core (:351-357)
    qkv.grad = None
    sim = torch.einsum("bhdi,bhdj->bhij", q * d ** -0.5, k)
    out = torch.einsum("bhij,bhdj->bhid", sim.softmax(dim=-1), v).permute(0, 1, 3, 2).reshape(b, heads * d, hw, hw)
    out.backward(dout)
    got = K.attention_bwd(nhwc(qkv), nhwc(dout), heads, d, d ** -0.5)
    assert_close(got.permute(0, 3, 1, 2), qkv.grad, TOL, "attention dqkv")


def test_small_ops_bwd():
    x = rnd(3, 5, 7, seed=11, scale=2.0).requires_grad_()
    dy = rnd(3, 5, 7, seed=12)
    for act, fn in ((K.GELU, F.gelu), (K.SILU, F.silu)):
        x.grad = None
        y = fn(x)
        y.backward(dy)
        assert_close(K.act_fwd(x.detach().to(DEV), act), y.detach(), 1e-6, "act fwd")
        assert_close(K.act_bwd(x.detach().to(DEV), dy.to(DEV), act), x.grad, TOL, "act bwd")
    u = rnd(2, 6, 4, 4, seed=13).requires_grad_()
    up = F.interpolate(u, scale_factor=2, mode="nearest")
    dup = rnd(2, 6, 8, 8, seed=14)
    up.backward(dup)
    assert_close(K.upsample2x_bwd(nhwc(dup)).permute(0, 3, 1, 2), u.grad, TOL, "upsample bwd")
    # loss: mean_b(mean(|pred-target|^p) * w[t])
    pred = rnd(3, 2, 4, 4, seed=15).requires_grad_()
    tgt = rnd(3, 2, 4, 4, seed=16)
    t = torch.tensor([1, 5, 2])
    w = torch.rand(8, generator=torch.Generator().manual_seed(17)) + 0.5
    for l1 in (False, True):
        pred.grad = None
        per = (pred - tgt).abs() if l1 else (pred - tgt) ** 2
        loss = (per.reshape(3, -1).mean(1) * w[t]).mean()
        (loss * 1.7).backward()
        got = K.diffusion_loss_bwd(nhwc(pred), nhwc(tgt), t.to(DEV), w.to(DEV), torch.tensor([1.7], device=DEV), l1)
        assert_close(got.permute(0, 3, 1, 2), pred.grad, TOL, f"loss bwd l1={l1}")
    # 7x7 weight gradient (the U-Net's first conv)
    xi = rnd(2, 6, 16, 16, seed=18).requires_grad_()
    wi = rnd(64, 6, 7, 7, seed=19, scale=0.1).requires_grad_()
    dyo = rnd(2, 64, 16, 16, seed=20)
    F.conv2d(xi, wi, padding=3).backward(dyo)
    assert_close(K.conv_wgrad(nhwc(xi), nhwc(dyo), 7), wi.grad, TOL, "7x7 wgrad")
    wt = K.conv_flip_weights(wi.detach().to(DEV).contiguous())
    assert_close(K.conv2d(nhwc(dyo), K.conv_pack(wt, K.F32), 6, 7, compute=K.F32, out_dtype=K.F32).permute(0, 3, 1, 2), xi.grad, TOL, "7x7 dgrad")


def build_unet(tag="m0", compute="f32"):
    from diffusion_prior import Unet
    dg = load_golden("diffusion.npz")
    m = meta_of(dg)[tag]
    u = Unet(dim=64, dim_mults=tuple(m["mults"]), channels=m["ch"], resnet_block_groups=m["groups"], compute=compute)
    sd = synth_sd(m["spec"], 21)
    u.load_state_dict(sd, strict=True)
    return u.to(DEV), sd, m


@pytest.mark.parametrize("tag", ["m0", "m2", "c5"])
def test_unet_and_p_losses_gradients_match_oracle_autograd(tag):
    """loss = p_losses(x0, t, eps) (gaussian_diffusion.py:823-859) for three U-Net geometries: every parameter gradient of
    the HIP training path against autograd through the CPU oracle on the same weights / inputs / noise."""
    from diffusion_prior import GaussianDiffusion
    u, sd, m = build_unet(tag)
    size, ch = m["size"], m["ch"]
    B = 2
    d = GaussianDiffusion(u, image_size=size, timesteps=1000, beta_schedule="cosine", loss_type="l2", auto_normalize=False).to(DEV)
    x0 = rnd(B, ch, size, size, seed=31, scale=0.8)
    eps = rnd(B, ch, size, size, seed=32)
    t = torch.tensor([10, 700])
    # oracle
    so = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    model = lambda x, tt: D.unet_forward(x, tt, so, 64, m["mults"], m["groups"])
    buf = D.diffusion_buffers("cosine", 1000)
    loss_ref = D.p_losses(buf, model, x0, t, eps, "l2")
    loss_ref.backward()
    # HIP training path
    u.train()
    loss = d.p_losses(x0.to(DEV), t.to(DEV), noise=eps.to(DEV))
    assert loss.requires_grad
    assert_close(loss, loss_ref.detach(), 1e-5, "loss")
    loss.backward()
    worst = ("", 0.0)
    for k, p in u.named_parameters():
        g_ref = so[k].grad
        assert p.grad is not None, f"no gradient for {k}"
        assert torch.isfinite(p.grad).all(), k
        e = relerr(p.grad, g_ref)
        worst = max(worst, (k, e), key=lambda kv: kv[1])
        assert e <= 2e-4, f"grad {k}: rel err {e:.3e}"
    print("worst gradient", worst)
    # Unet.forward in grad mode is the same graph
    y = u(x0.to(DEV), t.to(DEV))
    assert y.requires_grad and tuple(y.shape) == (B, ch, size, size)
    with torch.no_grad():
        assert_close(y, u(x0.to(DEV), t.to(DEV)), 1e-5, "training forward vs inference forward")


def test_diffusion_prior_training_step_runs():
    """The reference's prior-training loop body (diffusion_prior/trainer.py:134-150): frozen flow encodes, the prior's
    per-part losses are summed and optimised with Adam; the loss goes down."""
    from normalizing_flow import NFBackbone, initialize_with_zeros
    from diffusion_prior import DiffusionPrior, IdentityFormater
    nf = NFBackbone(model_dir="", in_channel=1, L=2, K=1, learn_prior_mean_logs=True, freeze_flow=True).to(DEV)
    fmt = IdentityFormater(2, 1, 16)
    shapes = fmt.get_input_shapes()
    prior = DiffusionPrior(latent_formater=fmt, unet_kwargs={"dim": 32, "dim_mults": (1, 2), "resnet_block_groups": 8},
                           diffusion_kwargs={"timesteps": 50, "beta_schedule": "cosine", "loss_type": "l2"},
                           image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes]).to(DEV)
    opt = torch.optim.Adam(prior.parameters(only_trainable=True), lr=2e-3)
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(8, 1, 16, 16, generator=g) - 0.5).to(DEV)
    nf.set_eval_mode()
    prior.train()
    with torch.no_grad():
        latents, _ = nf.transform(x, initialize_with_zeros(1, 8, torch.device(DEV)))
    torch.manual_seed(0)
    losses = []
    for _ in range(12):
        loss = sum(prior(latents))                                       # DiffusionPrior.forward -> list of part losses
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert np.mean(losses[-4:]) < np.mean(losses[:4]), losses


def test_joint_training_gradient_reaches_the_flow():
    """Unfrozen flow (diffusion_prior/trainer.py:139-141): loss = sum of the prior's part losses + 0.5 * bits/dim; the
    prior's loss must propagate through q_sample and the U-Net input into the flow's parameters.  Checked against autograd
    through the oracles on the same weights, inputs, timesteps and noise."""
    from oracle import nf_oracle as O
    from normalizing_flow import NFBackbone, initialize_with_zeros, calculate_loss
    from diffusion_prior import Unet, GaussianDiffusion
    from tests.golden.synth import synth_tensor
    nf = NFBackbone(model_dir="", in_channel=1, L=2, K=1, learn_prior_mean_logs=True, freeze_flow=False, compute="f32")
    fsd = {k: T(synth_tensor(k, v.shape, 61)) for k, v in nf.model.state_dict().items()}
    nf.model.load_state_dict(fsd)
    nf.to(DEV)
    u = Unet(dim=32, dim_mults=(1, 2), channels=8, resnet_block_groups=8, compute="f32")
    usd = {k: T(synth_tensor(k, v.shape, 62)) for k, v in u.state_dict().items()}
    u.load_state_dict(usd)
    d = GaussianDiffusion(u.to(DEV), image_size=4, timesteps=100, beta_schedule="cosine", loss_type="l2",
                          auto_normalize=False).to(DEV)
    B = 2
    x = rnd(B, 1, 16, 16, seed=63, scale=0.3)
    eps = rnd(B, 8, 4, 4, seed=64)
    t = torch.tensor([3, 77])
    # oracle: last latent part of the L=2 flow (8 x 4 x 4) into the prior's loss, plus 0.5 * bpd of the flow alone
    fo = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fsd.items()}
    uo = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in usd.items()}
    zs, ldj, _ = O.glow_fwd(x, fo, 2, 1, torch.zeros(B, dtype=torch.float64), None)
    buf = D.diffusion_buffers("cosine", 100)
    model = lambda xx, tt: D.unet_forward(xx, tt, uo, 32, [1, 2], 8)
    loss_ref = D.p_losses(buf, model, zs[-1], t, eps, "l2") + 0.5 * O.bits_per_dim(ldj, 32.0, 16 * 16 * 3.0)
    loss_ref.backward()
    # HIP
    nf.set_train_mode()
    u.train()
    lat, ll = nf.transform(x.to(DEV), initialize_with_zeros(1, B, torch.device(DEV)))
    assert lat[-1].requires_grad
    loss = d.p_losses(lat[-1], t.to(DEV), noise=eps.to(DEV)) + 0.5 * calculate_loss(ll, 32.0, 16 * 16 * 3.0)
    assert_close(loss, loss_ref.detach(), 1e-5, "joint loss")
    loss.backward()
    n_checked = 0
    for k, p in nf.model.named_parameters():
        if fo[k].grad is None:                      # Split priors are skipped on the NFBackbone path (logp=None): no gradient
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        if float(fo[k].grad.abs().max()) > 0:
            assert relerr(p.grad, fo[k].grad) <= 2e-4, f"flow grad {k}: {relerr(p.grad, fo[k].grad):.3e}"
            n_checked += 1
    assert n_checked >= 10
    for k, p in u.named_parameters():
        assert relerr(p.grad, uo[k].grad) <= 2e-4, f"unet grad {k}"


def test_captured_diffusion_training_step_matches_eager():
    """nfdp_hip.graphs.capture_step on the diffusion-prior training step (diffusion_prior/trainer.py:134-150 with the fused Adam): the
    captured step replays the eager step - same loss, same gradients, same parameters after each of three steps (timesteps and noise
    injected so both consume identical randomness).  Eager steps run on the DEFAULT stream before the capture on purpose: that order
    is what round 2 reported as a host segfault in hipStreamEndCapture (see tools/probe_capture_abort.py)."""
    from normalizing_flow import NFBackbone, initialize_with_zeros
    from diffusion_prior import DiffusionPrior, IdentityFormater
    from nfdp_hip.optim import FusedAdam
    from nfdp_hip.graphs import capture_step
    B = 8
    nf = NFBackbone(model_dir="", in_channel=1, L=2, K=1, learn_prior_mean_logs=True, freeze_flow=True).to(DEV)
    fmt = IdentityFormater(2, 1, 16)
    shapes = fmt.get_input_shapes()

    def make():
        torch.manual_seed(3)
        prior = DiffusionPrior(latent_formater=fmt, unet_kwargs={"dim": 32, "dim_mults": (1, 2), "resnet_block_groups": 8},
                               diffusion_kwargs={"timesteps": 50, "beta_schedule": "cosine", "loss_type": "l2"},
                               image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes]).to(DEV)
        g = torch.Generator().manual_seed(5)
        for p in prior.parameters():                                  # away from the zero-output init, identically for both copies
            p.data.add_(0.05 * torch.randn(p.shape, generator=g).to(DEV))
        prior.train()
        return prior, FusedAdam(list(prior.parameters(only_trainable=True)), lr=1e-3)

    g = torch.Generator().manual_seed(4)
    x = (torch.rand(B, 1, 16, 16, generator=g) - 0.5).to(DEV)
    with torch.no_grad():
        latents, _ = nf.transform(x, initialize_with_zeros(1, B, torch.device(DEV)))
    ts = [torch.randint(0, 50, (B,), generator=g).to(DEV) for _ in latents]
    eps = [torch.randn(z.shape, generator=g).to(DEV) for z in latents]

    def step_of(prior, opt):
        def step():
            loss = sum(pr.diffusion.p_losses(z, t, noise=e) for pr, z, t, e in zip(prior.priors, latents, ts, eps))
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss
        return step

    pe, oe = make()
    pc, oc = make()
    eager, cstep = step_of(pe, oe), step_of(pc, oc)
    for _ in range(2):                                                # eager steps on the legacy default stream first, for BOTH copies
        eager()
        cstep()
    cap = capture_step(cstep, warmup=1)                               # one more (side-stream) warm-up step; the capture itself executes nothing
    eager()                                                           # ... so the eager copy has taken as many steps: 2 + 1
    with pytest.raises(RuntimeError):
        capture_step(cstep, warmup=0)
    for k in range(3):
        l_e = eager()
        l_c = cap()
        torch.cuda.synchronize()
        assert_close(l_c, l_e.detach(), 1e-5, f"replay {k}: loss")
        assert relerr(oc.flat_grad, oe.flat_grad) <= 1e-4, f"replay {k}: gradient {relerr(oc.flat_grad, oe.flat_grad):.2e}"
        assert relerr(oc.flat_param, oe.flat_param) <= 1e-5, f"replay {k}: parameters"
    assert oc.step_count == oe.step_count == 6
