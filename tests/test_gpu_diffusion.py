"""GPU parity of the diffusion-prior path (U-Net eps-predictor, Gaussian diffusion steps/loops, composition) against
golden vectors captured from the reference (tests/golden/diffusion.npz) with all randomness injected.
fp32 mode: 1e-4 relative (BASELINE.json gate); bf16 MFMA mode at a separately stated looser tolerance."""
import pytest
import torch

import nfdp_hip as K
from tests.util import load_golden, meta_of, synth_sd, T, assert_close, relerr

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


@pytest.fixture(scope="module")
def dg():
    return load_golden("diffusion.npz")


def make_unet(m, compute="f32"):
    from diffusion_prior import Unet
    u = Unet(dim=64, dim_mults=tuple(m["mults"]), channels=m["ch"], resnet_block_groups=m["groups"], compute=compute)
    u.load_state_dict(synth_sd(m["spec"], 21), strict=True)
    return u.to(DEV)


@pytest.mark.parametrize("tag", ["m0", "m2", "c0", "c5"])
@torch.no_grad()
def test_unet_forward_golden(dg, tag):
    m = meta_of(dg)[tag]
    x, t = T(dg[f"unet{tag}_x"]).to(DEV), T(dg[f"unet{tag}_t"]).to(DEV)
    y = make_unet(m)(x, t)
    assert tuple(y.shape) == dg[f"unet{tag}_y"].shape
    assert_close(y, dg[f"unet{tag}_y"], TOL, "unet f32")
    yb = make_unet(m, "bf16")(x, t)
    assert_close(yb, dg[f"unet{tag}_y"], 3e-2, "unet bf16")


@torch.no_grad()
def test_unet_celeba_part_golden():
    """The U-Net of BASELINE configs[4]'s first latent part (6x64x64, dim_mults [1,2,2], groups 16) against the REFERENCE's outputs
    (tests/golden/unet_celeba.npz): the chunked linear-attention core (n = 4096), LDS-staged 64x64 GroupNorm groups, full attention
    at n = 256; forward, p_losses and one p_sample with the recorded noise."""
    from diffusion_prior import GaussianDiffusion
    cg = load_golden("unet_celeba.npz")
    m = meta_of(cg)
    x, t, eps = T(cg["x"]).to(DEV), T(cg["t"]).to(DEV), T(cg["eps"]).to(DEV)

    def unet(compute):
        from diffusion_prior import Unet
        u = Unet(dim=64, dim_mults=tuple(m["mults"]), channels=m["ch"], resnet_block_groups=m["groups"], compute=compute)
        u.load_state_dict(synth_sd(m["spec"], 27), strict=True)
        return u.to(DEV)

    u = unet("f32")
    assert_close(u(x, t), cg["y"], TOL, "unet 6x64x64 f32")
    assert_close(unet("bf16")(x, t), cg["y"], 6e-2, "unet 6x64x64 bf16")      # bf16 convs, element-wise: observed 3.1e-2
    d = GaussianDiffusion(u, image_size=64, timesteps=1000, beta_schedule="cosine", loss_type="l2", auto_normalize=False).to(DEV)
    assert_close(d.p_losses(x * 0.8, t, noise=eps), cg["p_losses"], TOL, "p_losses 6x64x64")
    img, x0 = d.p_sample(x * 0.8, int(t[0]), noise=T(cg["psample_noise"]).to(DEV))
    assert_close(img, cg["psample_img"], TOL, "p_sample img 6x64x64")
    assert_close(x0, cg["psample_x0"], TOL, "p_sample x0 6x64x64")


@torch.no_grad()
def test_config3_mnist_diffusion_prior_end_to_end():
    """BASELINE configs[3] as one chain: frozen Glow L3 K4 on 1x32x32 encodes -> three dim-64 U-Net priors -> T = 1000 ancestral
    reverse chain per latent part (hipGraph-captured step, parts interleaved on their own streams) -> Glow.sample decodes.
    Untrained weights, so the check is finiteness, shapes and bitwise determinism under a fixed seed; the single steps and short
    loops are pinned against the reference in the tests above."""
    from normalizing_flow import NFBackbone
    from diffusion_prior import DiffusionPrior, IdentityFormater
    from tests.golden.synth import synth_tensor
    B = 4
    nf = NFBackbone(None, in_channel=1, L=3, K=4, learn_prior_mean_logs=True, freeze_flow=True)
    nf.model.load_state_dict({k: T(synth_tensor(k, v.shape, 1)) for k, v in nf.model.state_dict().items()}, strict=True)
    fmt = IdentityFormater(3, 1, 32)
    shapes = fmt.get_input_shapes()
    prior = DiffusionPrior(latent_formater=fmt, unet_kwargs=dict(dim=64, dim_mults=(1, 2), resnet_block_groups=8),
                           diffusion_kwargs=dict(timesteps=1000, sampling_timesteps=1000, loss_type="l2", beta_schedule="cosine"),
                           image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes])
    g = torch.Generator().manual_seed(8)
    for p in prior.parameters():                                    # away from the zero-output init
        p.data.add_(0.02 * torch.randn(p.shape, generator=g).to(DEV))
    x = torch.rand(B, 1, 32, 32, generator=g).to(DEV) - 0.5
    lat, ldj = nf.transform(x, torch.zeros(B, dtype=torch.float64, device=DEV))
    assert torch.isfinite(ldj).all()
    losses = prior(lat)
    assert len(losses) == 3 and all(torch.isfinite(l) for l in losses)
    prior.sample_latents(B)                                         # captures the step graphs
    runs = []
    for _ in range(2):
        torch.manual_seed(2024)
        zs = prior.sample_latents(B)
        img = nf.sample(zs)
        torch.cuda.synchronize()
        runs.append([z.clone() for z in zs] + [img.clone()])
    assert [tuple(z.shape) for z in runs[0][:3]] == [(B, *[int(v) for v in s]) for s in shapes]
    assert runs[0][3].shape == (B, 1, 32, 32)
    for a, b in zip(*runs):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    assert all(float(z.abs().max()) <= 1.0 + 1e-6 for z in runs[0][:3])      # x0 clamp of the last step (:700-701)


def make_diffusion(dg, timesteps=1000, **kw):
    from diffusion_prior import GaussianDiffusion
    u = make_unet(meta_of(dg)["m0"])
    return GaussianDiffusion(u, image_size=16, timesteps=timesteps, beta_schedule="cosine", loss_type="l2",
                             auto_normalize=False, **kw).to(DEV)


@torch.no_grad()
def test_schedule_buffers_golden(dg):
    from diffusion_prior import GaussianDiffusion
    from oracle.diffusion_oracle import BUFFER_NAMES
    u = make_unet(meta_of(dg)["m0"])
    for sched in ("cosine", "linear", "sigmoid"):
        d = GaussianDiffusion(u, image_size=16, timesteps=1000, beta_schedule=sched, auto_normalize=False)
        for name in BUFFER_NAMES:
            assert_close(getattr(d, name), dg[f"buf_{sched}_{name}"], 1e-6, name)
        helpers = ["_neg_sqrt_recipm1", "_x0_a", "_x0_b", "_neg_sqrt_one_minus_ac"]       # kernel-side tables, not part of the state
        assert [k for k, _ in d.named_buffers(recurse=False)] == list(BUFFER_NAMES) + helpers
        assert not any(h in d.state_dict() for h in helpers)
    with pytest.raises(ValueError):
        GaussianDiffusion(u, image_size=16, beta_schedule="nope", auto_normalize=False)


@torch.no_grad()
def test_diffusion_process_golden(dg):
    d = make_diffusion(dg)
    x0, eps, t = T(dg["d_x0"]).to(DEV), T(dg["d_eps"]).to(DEV), T(dg["d_t"]).to(DEV)
    assert_close(d.q_sample(x0, t, eps), dg["d_q_sample"], TOL, "q_sample")
    assert_close(d.predict_start_from_noise(x0, t, eps), dg["d_pred_x0"], TOL, "pred_x0")
    pm, pv, plv = d.q_posterior(x0, eps, t)
    assert_close(pm, dg["d_post_mean"], TOL, "post mean")
    assert_close(pv, dg["d_post_var"], TOL, "post var")
    assert_close(plv, dg["d_post_logvar"], TOL, "post logvar")
    assert_close(d.p_losses(x0, t, noise=eps), dg["d_p_losses_l2"], TOL, "l2")
    d.loss_type = "l1"
    assert_close(d.p_losses(x0, t, noise=eps), dg["d_p_losses_l1"], TOL, "l1")
    d.loss_type = "l2"
    assert_close(d.p_losses(x0, T(dg["d_fwd_t"]).to(DEV), noise=T(dg["d_fwd_noise"]).to(DEV)), dg["d_fwd_loss"], TOL, "fwd")
    assert torch.isfinite(d(x0))                                   # forward(): t from the device RNG
    d.loss_type = "huber"
    with pytest.raises(ValueError):
        d.p_losses(x0, t, noise=eps)
    with pytest.raises(AssertionError):
        d(torch.zeros(1, 2, 8, 8, device=DEV))                     # wrong image size (reference :863)


@pytest.mark.parametrize("step", [0, 1, 500, 999])
@torch.no_grad()
def test_p_sample_golden(dg, step):
    d = make_diffusion(dg)
    x = T(dg["d_x0"]).to(DEV)
    img, x0 = d.p_sample(x, step, noise=T(dg[f"d_psample{step}_noise"]).to(DEV))
    assert_close(img, dg[f"d_psample{step}_img"], TOL, "img")
    assert_close(x0, dg[f"d_psample{step}_x0"], TOL, "x0")
    assert float(x0.abs().max()) <= 1.0                            # x0 clamp kept although latents are not normalised
    mp = d.model_predictions(x, torch.full((2,), step, device=DEV, dtype=torch.long), clip_x_start=True)
    assert_close(mp.pred_x_start, dg[f"d_psample{step}_x0"], TOL, "model_predictions")


@torch.no_grad()
def test_sampling_loops_golden(dg):
    d8 = make_diffusion(dg, timesteps=8)
    out = d8.p_sample_loop((2, 2, 16, 16), x_T=T(dg["loop8_xT"]), noises=list(T(dg["loop8_noises"])))
    assert_close(out, dg["loop8_out"], TOL, "p_sample_loop T=8")
    dd = make_diffusion(dg, timesteps=8, sampling_timesteps=4, ddim_sampling_eta=1.0)
    assert dd.is_ddim_sampling
    out = dd.ddim_sample((2, 2, 16, 16), x_T=T(dg["ddim_xT"]), noises=list(T(dg["ddim_noises"])))
    assert_close(out, dg["ddim_out"], TOL, "ddim T=8 S=4")
    assert d8.use_graph
    s = d8.sample(batch_size=3)                                    # device RNG path, hipGraph-captured step
    assert ("step", (3, 16, 16, 2)) in [k[:2] for k in d8._graphs]
    d8.use_graph = False
    s2 = d8.sample(batch_size=3)                                   # eager loop
    assert s2.shape == s.shape and torch.isfinite(s2).all()
    d8.use_graph = True
    assert s.shape == (3, 2, 16, 16) and torch.isfinite(s).all()
    allt = d8.sample(batch_size=2, return_all_timesteps=True)
    assert allt.shape == (2, 9, 2, 16, 16)


@torch.no_grad()
def test_diffusion_prior_composition_with_glow():
    """run_diffusion_prior_experiment.py wiring: frozen Glow encodes, one diffusion model per latent part."""
    from normalizing_flow import NFBackbone, calculate_output_shapes
    from diffusion_prior import DiffusionPrior, IdentityFormater
    nf = NFBackbone(None, in_channel=1, L=3, K=1, learn_prior_mean_logs=True, freeze_flow=True)
    fmt = IdentityFormater(3, 1, 32)
    shapes = fmt.get_input_shapes()
    assert [tuple(s) for s in shapes] == calculate_output_shapes(3, 1, 32)
    prior = DiffusionPrior(latent_formater=fmt, unet_kwargs=dict(dim=64, dim_mults=(1, 2), resnet_block_groups=8),
                           diffusion_kwargs=dict(timesteps=4, sampling_timesteps=4, loss_type="l2", beta_schedule="cosine"),
                           image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes])
    assert len(prior.state_dict()) > 0                             # the reference's list-held priors have 0 keys
    assert sum(p.numel() for p in prior.parameters()) > 9e6
    x = torch.rand(4, 1, 32, 32, device=DEV) - 0.5
    lat, _ = nf.transform(x, torch.zeros(4, dtype=torch.float64, device=DEV))
    losses = prior(lat)
    assert len(losses) == 3 and all(torch.isfinite(l) for l in losses)
    zs = prior.sample_latents(2)
    assert [tuple(z.shape[1:]) for z in zs] == [tuple(int(v) for v in s) for s in shapes]
    img = nf.sample(zs)
    assert img.shape == (2, 1, 32, 32) and torch.isfinite(img).all()


@torch.no_grad()
def test_concurrent_part_streams_match_sequential_sampling():
    """The per-part reverse chains run on one HIP stream each; the host enqueues them in the reference's order, so with the
    same seed the samples are bit-identical to the sequential loop (p_sample_loop through the captured step, and
    sample_latent_given_start through the eager step)."""
    from diffusion_prior import DiffusionPrior, IdentityFormater
    fmt = IdentityFormater(3, 1, 32)
    shapes = fmt.get_input_shapes()
    prior = DiffusionPrior(latent_formater=fmt, unet_kwargs=dict(dim=32, dim_mults=(1, 2), resnet_block_groups=8),
                           diffusion_kwargs=dict(timesteps=6, sampling_timesteps=6, loss_type="l2", beta_schedule="cosine"),
                           image_sizes=[int(s[1]) for s in shapes], latent_channels=[int(s[0]) for s in shapes])
    g = torch.Generator().manual_seed(5)
    for p in prior.parameters():                                    # away from the zero-output init
        p.data.add_(0.05 * torch.randn(p.shape, generator=g).to(DEV))
    prior.sample_latents(3)                                         # captures the step graphs (the capture itself draws from the RNG)
    outs = {}
    for mode in (False, True, True):
        prior.concurrent_parts = mode
        torch.manual_seed(11)
        zs = prior.sample_latents(3)
        starts = [torch.randn(3, *[int(v) for v in s], device=DEV) for s in shapes]
        gs = prior.sample_latents_given_start(starts)
        torch.cuda.synchronize()
        outs.setdefault(mode, []).append([z.clone() for z in zs] + [z.clone() for z in gs])
    for other in outs[True]:
        for a, b in zip(outs[False][0], other):
            assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("hw,b", [(16, 3), (20, 2), (32, 2), (64, 1)])
def test_linear_attention_core_long_sequences(hw, b):
    """LinearAttention core (gaussian_diffusion.py:320-332) at n = 256 (one workgroup per head) and beyond (positions chunked over
    the grid, streaming-softmax recombination; n = 400 has a ragged last chunk) against the plain formula."""
    heads, d, n = 4, 32, hw * hw
    g = torch.Generator().manual_seed(hw)
    qkv = torch.randn(b, 3 * heads * d, hw, hw, generator=g) * 1.5
    q, k, v = (t.reshape(b, heads, d, n) for t in qkv.chunk(3, dim=1))
    ctx = torch.einsum("bhdn,bhen->bhde", k.softmax(dim=-1), v / n)
    ref = torch.einsum("bhde,bhdn->bhen", ctx, q.softmax(dim=-2) * d ** -0.5).reshape(b, heads * d, hw, hw)
    got = K.linear_attention(qkv.permute(0, 2, 3, 1).contiguous().to(DEV), heads, d, d ** -0.5)
    err = float((got.permute(0, 3, 1, 2).cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


@pytest.mark.parametrize("hw,b", [(8, 3), (16, 2), (20, 2), (32, 1)])
def test_full_attention_core_long_sequences(hw, b):
    """Attention core (gaussian_diffusion.py:351-357) at n <= 256 (keys in LDS once) and beyond (queries chunked over the grid,
    keys streamed) against the plain formula."""
    heads, d, n = 4, 32, hw * hw
    g = torch.Generator().manual_seed(100 + hw)
    qkv = torch.randn(b, 3 * heads * d, hw, hw, generator=g)
    q, k, v = (t.reshape(b, heads, d, n) for t in qkv.chunk(3, dim=1))
    sim = torch.einsum("bhdi,bhdj->bhij", q * d ** -0.5, k)
    ref = torch.einsum("bhij,bhdj->bhid", sim.softmax(dim=-1), v).permute(0, 1, 3, 2).reshape(b, heads * d, hw, hw)
    got = K.attention(qkv.permute(0, 2, 3, 1).contiguous().to(DEV), heads, d, d ** -0.5)
    err = float((got.permute(0, 3, 1, 2).cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


@pytest.mark.parametrize("hw,c,groups", [(8, 64, 8), (32, 64, 16), (64, 64, 16), (64, 192, 8), (48, 66, 3)])
def test_groupnorm_silu_all_group_sizes(hw, c, groups):
    """Block's GroupNorm -> (scale+1) x + shift -> SiLU (gaussian_diffusion.py:267-276) for groups that fit the registers, that
    are staged in LDS (32x32 / 64x64 latent parts) and that take the three-pass global path (too large or odd widths)."""
    import torch.nn.functional as F
    b = 2
    g = torch.Generator().manual_seed(hw + c)
    x = torch.randn(b, c, hw, hw, generator=g) * 2.0 + 0.3
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    ss = torch.randn(b, 2 * c, generator=g) * 0.5
    ref = F.group_norm(x, groups, gamma, beta, eps=1e-5)
    ref = F.silu(ref * (ss[:, :c, None, None] + 1) + ss[:, c:, None, None])
    got = K.groupnorm_silu(x.permute(0, 2, 3, 1).contiguous().to(DEV), groups, gamma.to(DEV), beta.to(DEV), ss.to(DEV), 1e-5)
    err = float((got.permute(0, 3, 1, 2).cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


@torch.no_grad()
def test_ddim_captured_loop_matches_eager_loop(dg):
    """ddim_sample on the captured step (scalars and timesteps from device tables, device-side step counter, per-graph Philox
    generator) against the eager loop fed the very same noise: the generator is re-created here from the same seed draw."""
    dd = make_diffusion(dg, timesteps=50, sampling_timesteps=7, ddim_sampling_eta=1.0)
    assert dd.is_ddim_sampling
    shape = (3, 2, 16, 16)
    torch.manual_seed(123)
    got = dd.ddim_sample(shape)                                   # captures on first use (the capture draws from its own generator)
    torch.manual_seed(123)
    got = dd.ddim_sample(shape)
    torch.manual_seed(123)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    g2 = torch.Generator(device=DEV).manual_seed(seed)
    x_T = torch.empty(3, 16, 16, 2, device=DEV).normal_(generator=g2)
    noises = [torch.randn(3, 16, 16, 2, device=DEV, generator=g2) for _ in range(6)]
    dd.use_graph = False
    ref = dd.ddim_sample(shape, x_T=x_T.permute(0, 3, 1, 2), noises=[n.permute(0, 3, 1, 2) for n in noises])
    assert torch.isfinite(got).all()
    assert_close(got, ref, 1e-5, "captured vs eager DDIM")


@torch.no_grad()
def test_interpolate_and_given_start_on_the_captured_chain(dg):
    """GaussianDiffusion.interpolate (:785-804) and DiffusionModel.sample_latent_given_start (model.py:57-70): the reverse half on
    the captured step equals the eager p_sample loop fed the same noise (the chain's generator is re-created from the seed draw)."""
    d = make_diffusion(dg, timesteps=9)
    shape = (2, 2, 16, 16)
    g = torch.Generator().manual_seed(4)
    x1, x2 = torch.randn(shape, generator=g).to(DEV), torch.randn(shape, generator=g).to(DEV)
    d.interpolate(x1, x2, t=6, lam=0.3)                           # captures the step
    outs = {}
    for mode in (True, False):
        d.use_graph = mode
        torch.manual_seed(77)
        if mode:
            outs[mode] = d.interpolate(x1, x2, t=6, lam=0.3)
        else:                                                      # the eager loop with the captured chain's noise
            tb = torch.full((2,), 6, device=DEV, dtype=torch.long)
            xt1, xt2 = d.q_sample(x1, t=tb), d.q_sample(x2, t=tb)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            g2 = torch.Generator(device=DEV).manual_seed(seed)
            img = K.to_nhwc(0.7 * xt1 + 0.3 * xt2).contiguous()
            for i in reversed(range(0, 6)):
                nz = torch.randn(img.shape, device=DEV, generator=g2) if i > 0 else None
                img, _ = d._p_sample_nhwc(img, i, nz, False)
            outs[mode] = K.as_nchw(img)
    assert torch.isfinite(outs[True]).all()
    assert_close(outs[True], outs[False], 1e-5, "interpolate: captured vs eager chain")


@torch.no_grad()
def test_captured_sampling_graph_follows_weight_updates(dg):
    """Sampling between training steps (diffusion_prior/trainer.py:181-182): the captured reverse step must not keep using the
    weights it was captured with.  After an in-place weight update the same seed gives different samples, equal to those of the
    eager loop with the new weights."""
    d = make_diffusion(dg, timesteps=6)
    shape = (2, 2, 16, 16)
    torch.manual_seed(5)
    d.sample(batch_size=2)                                        # capture
    torch.manual_seed(5)
    s_old = d.sample(batch_size=2)
    for p in d.model.parameters():
        p.mul_(1.05)                                              # what an optimizer step does (bumps the tensor versions)
    torch.manual_seed(5)
    s_new = d.sample(batch_size=2)
    assert not torch.equal(s_old, s_new)
    torch.manual_seed(5)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    g2 = torch.Generator(device=DEV).manual_seed(seed)
    x_T = torch.empty(2, 16, 16, 2, device=DEV).normal_(generator=g2)
    noises = [torch.randn(2, 16, 16, 2, device=DEV, generator=g2) for _ in range(5)]
    d.use_graph = False
    ref = d.p_sample_loop(shape, x_T=x_T.permute(0, 3, 1, 2), noises=[n.permute(0, 3, 1, 2) for n in noises] + [None])
    assert_close(s_new, ref, 1e-5, "captured loop after a weight update vs eager loop")


@torch.no_grad()
def test_time_path_table_equals_the_direct_time_path(dg):
    """Unet._scale_shifts_for: the [T, sum 2*dim_out] table gathered at `time` against the four launches it replaces (sinusoidal
    embedding, two Linears, the blocks' projections), and the U-Net output through either; the table follows a weight update."""
    m = meta_of(dg)["m0"]
    u = make_unet(m)
    x, t = T(dg["unetm0_x"]).to(DEV), torch.tensor([3, 977], device=DEV)
    direct = u._all_scale_shifts(u.time_embedding(t, silu_out=True))
    assert u._time_table(x.device) is None                          # a bare Unet has no table ...
    y_direct = u(x, t)
    u.enable_time_table(1000)                                       # ... GaussianDiffusion enables it with its T
    tab = u._scale_shifts_for(t)
    for k in direct:
        assert_close(tab[k], direct[k], 1e-6, "scale/shift rows")
    assert_close(u(x, t), y_direct, 1e-6, "U-Net output through the table")
    assert_close(u(x, t), dg["unetm0_y"], TOL, "U-Net output through the table vs the reference")
    key0 = u._tt_key
    u.time_mlp[1].weight.mul_(1.01)                                 # what an optimizer step does
    after = u._scale_shifts_for(t)
    assert u._tt_key != key0
    ref = u._all_scale_shifts(u.time_embedding(t, silu_out=True))
    for k in ref:
        assert_close(after[k], ref[k], 1e-6, "rows after a weight update")
    with torch.enable_grad():
        assert u._time_table(x.device) is None                      # the differentiable path computes the time MLP (it has gradients)


@pytest.mark.parametrize("b,cin,cout,hw,groups,with_ss,with_res", [(260, 64, 64, 8, 8, True, False), (7, 128, 128, 4, 8, True, True),
                                                                      (3, 192, 128, 2, 8, False, True), (130, 64, 128, 8, 16, True, True),
                                                                      (64, 128, 128, 4, 8, True, True), (33, 128, 64, 4, 8, True, False)])
@torch.no_grad()
def test_block_in_one_launch_matches_conv_then_groupnorm(b, cin, cout, hw, groups, with_ss, with_res):
    """Block (WS-conv -> GroupNorm -> scale/shift -> SiLU -> + residual, gaussian_diffusion.py:267-276, :301) as ONE launch
    (nfdp_conv2d_gn_silu_nhwc_f32: GroupNorm in the split-K conv's epilogue) against the two launches it replaces and against torch,
    for whole-image pixel tiles of 64 / 32 / 16 pixels, groups of 4..16 channels, ragged batches."""
    import torch.nn.functional as F
    from diffusion_prior.gaussian_diffusion import Block
    g = torch.Generator().manual_seed(b * 100 + hw)
    blk = Block(cin, cout, groups=groups).to(DEV)
    for p in blk.parameters():
        p.data.copy_((torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 1 else 0.08)).to(DEV))
    blk.norm.weight.data.add_(1.0)
    x = torch.randn(b, hw, hw, cin, generator=g).to(DEV)
    ss = (torch.randn(b, 2 * cout + 8, generator=g) * 0.5).to(DEV)[:, 4:4 + 2 * cout] if with_ss else None      # a column slice, as the U-Net passes
    res = torch.randn(b, hw, hw, cout, generator=g).to(DEV) if with_res else None
    assert K.conv2d_gn_silu_supported(x, cout, 3, groups, blk.proj.bias, blk.norm.weight, blk.norm.bias)
    K.start_trace()
    fused = blk.forward_nhwc(x, ss, "f32", residual=res)
    labels = [e["label"] for e in K.stop_trace()]
    assert any(lb.startswith("conv_splitk_f32+gn") for lb in labels) and not any(lb.startswith("groupnorm") for lb in labels), labels
    K.set_conv_gn_fused(False)
    try:
        plain = blk.forward_nhwc(x, ss, "f32", residual=res)
    finally:
        K.set_conv_gn_fused(True)
    assert_close(fused, plain, 2e-5, "one launch vs conv + GroupNorm launches")
    # torch reference of the same Block
    w = blk.proj.weight.detach().cpu()
    mean, var = w.mean(dim=(1, 2, 3), keepdim=True), w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    wn = (w - mean) * (var + 1e-5).rsqrt()
    h = F.conv2d(x.cpu().permute(0, 3, 1, 2), wn, blk.proj.bias.detach().cpu(), padding=1)
    h = F.group_norm(h, groups, blk.norm.weight.detach().cpu(), blk.norm.bias.detach().cpu(), eps=1e-5)
    if with_ss:
        sc = ss.cpu()
        h = h * (sc[:, :cout, None, None] + 1) + sc[:, cout:, None, None]
    h = F.silu(h)
    if with_res:
        h = h + res.cpu().permute(0, 3, 1, 2)
    assert_close(fused.permute(0, 3, 1, 2), h, 5e-5, "one-launch Block vs torch")
    assert not K.conv2d_gn_silu_supported(torch.zeros(2, 16, 16, cin, device=DEV), cout, 3, groups)       # 256-pixel images: two launches
    assert not K.conv2d_gn_silu_supported(torch.zeros(8, 8, 8, cin, device=DEV), cout, 3, groups)         # few 64-pixel images: the plain conv's smaller tiles win


@pytest.mark.parametrize("b,c1,c2,cout,hw,k", [(64, 64, 64, 64, 16, 3), (64, 128, 64, 128, 8, 3), (64, 128, 128, 128, 4, 3),
                                              (64, 64, 64, 64, 16, 1), (5, 128, 64, 128, 8, 1), (3, 64, 64, 64, 6, 3)])
@torch.no_grad()
def test_two_source_conv_equals_conv_of_the_concatenation(b, c1, c2, cout, hw, k):
    """nfdp_conv2d_cat_nhwc_f32: conv2d(torch.cat((x1, x2), dim=1)) read from the two tensors (the U-Net's skip connections,
    gaussian_diffusion.py:479-486) - the same kernels walking the same K order, so identical bits to the conv of the materialised
    concatenation; also from channel slices of wider tensors, and against torch."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(c1 + hw + k)
    wide = torch.randn(b, hw, hw, c1 + 8, generator=g).to(DEV)
    x1 = wide[..., :c1]                                              # a pixel-strided view
    x2 = torch.randn(b, hw, hw, c2, generator=g).to(DEV)
    w = (torch.randn(cout, c1 + c2, k, k, generator=g) * 0.05).to(DEV)
    bias = torch.randn(cout, generator=g).to(DEV)
    packed = K.conv_pack(w, K.F32)
    assert K.cat_sources_ok(x1, x2)
    two = K.conv2d_cat(x1, x2, packed, cout, k, add=bias)
    one = K.conv2d(K.concat_channels(x1.contiguous(), x2), packed, cout, k, add=bias)
    assert torch.equal(two, one)
    ref = F.conv2d(torch.cat((x1, x2), dim=3).permute(0, 3, 1, 2).cpu().double(), w.cpu().double(), bias.cpu().double(), padding=k // 2)
    assert_close(two.permute(0, 3, 1, 2), ref, 2e-5, "two-source conv vs torch fp64")


@torch.no_grad()
def test_unet_reads_skip_connections_from_both_tensors():
    """Unet.forward_nhwc in exact fp32: no concat launch is left (every up-path ResnetBlock's 3x3 and 1x1 convs read (x, skip) as two
    sources, the one-launch Blocks of the 8x8 / 4x4 stages included) and the output equals the run that materialises the
    concatenations bit for bit, for the three latent-part geometries of the 3x32x32 model."""
    from diffusion_prior import Unet
    for ch, size in ((6, 16), (12, 8), (48, 4)):
        torch.manual_seed(ch)
        u = Unet(dim=64, dim_mults=(1, 2), channels=ch, resnet_block_groups=8).to(DEV)
        x = torch.randn(8, size, size, ch, device=DEV)
        t = torch.randint(0, 1000, (8,), device=DEV)
        K.start_trace()
        a = u.forward_nhwc(x, t)
        labels = [e["label"] for e in K.stop_trace()]
        assert not any("concat" in lb for lb in labels), labels
        # 5 ResnetBlocks x (3x3 conv + 1x1 res_conv), the two in one launch where the res_conv rides in block1's conv
        assert sum(2 if "+res" in lb else 1 for lb in labels if "cat" in lb) == 10, labels
        K.set_conv_cat(False)
        try:
            K.start_trace()
            bref = u.forward_nhwc(x, t)
            labels = [e["label"] for e in K.stop_trace()]
        finally:
            K.set_conv_cat(True)
        assert sum("concat" in lb for lb in labels) == 5
        assert torch.equal(a, bref), f"{ch}x{size}x{size}"


@pytest.mark.parametrize("b,hw,c1,c2,cout,gn", [(64, 16, 64, 0, 64, False), (64, 16, 64, 64, 64, False), (64, 8, 128, 0, 128, False),
                                                 (64, 8, 128, 64, 128, False), (64, 4, 128, 0, 128, False), (3, 6, 64, 0, 100, False),
                                                 (5, 16, 32, 0, 64, False), (130, 8, 64, 0, 128, True), (64, 4, 128, 128, 128, True),
                                                 (7, 2, 192, 0, 128, True), (2, 40, 64, 0, 64, False), (8, 32, 64, 0, 64, False),
                                                 (4, 32, 64, 64, 64, False)])
@torch.no_grad()
def test_lds_staged_split_product_conv_is_bit_identical_to_the_gathering_kernel(b, hw, c1, c2, cout, gn):
    """conv3x3_ldsx_kernel (input tile + halo loaded and split once into LDS) against conv_splitk_f32_kernel<EMU> (every element gathered
    and split per tap): same stage order, same six products, same partial sums - identical bits; 64- / 32- / 16-pixel tiles, two-source
    inputs, ragged last tiles, images wider than a tile, the Block epilogue; and against torch fp64 (gaussian_diffusion.py:267-276)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(b + hw + c1 + c2 + cout)
    wide = torch.randn(b, hw, hw, c1 + 8, generator=g).to(DEV)
    x1 = wide[..., 4:4 + c1]                                         # a pixel-strided, 16-byte aligned view
    x2 = torch.randn(b, hw, hw, c2, generator=g).to(DEV) if c2 else None
    w = (torch.randn(cout, c1 + c2, 3, 3, generator=g) * 0.05).to(DEV)
    bias = torch.randn(cout, generator=g).to(DEV)
    packed = K.conv_pack(w, K.F32)
    assert K.get_f32_products() == "bf16x6"
    setter = K.lib().nfdp_debug_set_conv_ldsx
    if gn:
        gamma, beta = (torch.randn(cout, generator=g) * 0.3 + 1).to(DEV), (torch.randn(cout, generator=g) * 0.3).to(DEV)
        ss = (torch.randn(b, 2 * cout, generator=g) * 0.5).to(DEV)
        res = torch.randn(b, hw, hw, cout, generator=g).to(DEV)
        assert K.conv2d_gn_silu_supported(x1, cout, 3, 8, bias, gamma, beta, extra_cin=c2)
        run = lambda: K.conv2d_gn_silu(x1, packed, cout, 3, bias, 8, gamma, beta, ss, residual=res, x2=x2)
    elif c2:
        run = lambda: K.conv2d_cat(x1, x2, packed, cout, 3, add=bias)
    else:
        run = lambda: K.conv2d(x1, packed, cout, 3, add=bias)
    K.start_trace()
    new = run()
    labels = [e["label"] for e in K.stop_trace()]
    assert any("splitk" in (lb or "") or "conv2d_cat" in (lb or "") for lb in labels), labels
    setter(0)
    try:
        old = run()
    finally:
        setter(1)
    assert torch.equal(new, old)
    if not gn:
        xin = torch.cat((x1, x2), dim=3) if c2 else x1
        ref = F.conv2d(xin.permute(0, 3, 1, 2).cpu().double(), w.cpu().double(), bias.cpu().double(), padding=1)
        assert_close(new.permute(0, 3, 1, 2), ref, 2e-5, "LDS-staged conv vs torch fp64")


@pytest.mark.parametrize("b,hw,dim,kind", [(64, 16, 64, "linear"), (64, 8, 128, "linear"), (5, 6, 64, "linear"), (64, 8, 128, "full"),
                                           (3, 4, 128, "full"), (7, 4, 96, "linear"), (3, 32, 64, "linear"), (2, 64, 128, "linear"), (64, 4, 64, "linear"),
                                           (64, 2, 128, "linear"), (64, 2, 128, "full"), (5, 6, 128, "full")])
@torch.no_grad()
def test_attention_block_with_layernorms_inside_the_1x1_convs(b, hw, dim, kind):
    """Residual(PreNorm(dim, LinearAttention | Attention)) with the channel LayerNorms inside the 1x1 convs' launches
    (nfdp_conv1x1_ln_f32: LayerNorm of the input in to_qkv's launch; LayerNorm of the output + residual in to_out's) against the
    five-launch form (same LayerNorm arithmetic, the convs' products in another K association: 2e-5) and against torch; ragged pixel
    tiles, 64 / 96 / 128 channels."""
    from diffusion_prior.gaussian_diffusion import LinearAttention, Attention, PreNorm, Residual, _attn_block_nhwc
    torch.manual_seed(b + hw + dim)
    att = (LinearAttention if kind == "linear" else Attention)(dim)
    blk = Residual(PreNorm(dim, att)).to(DEV)
    for p_ in blk.parameters():
        p_.data.copy_(torch.randn_like(p_) * (0.2 if p_.dim() > 1 else 0.3) + (1.0 if p_.dim() == 4 and p_.shape[0] == 1 else 0.0))
    x = torch.randn(b, hw, hw, dim, device=DEV)
    K.start_trace()
    fused = _attn_block_nhwc(blk, x, "f32")
    labels = [e["label"] for e in K.stop_trace()]
    # (full attention: to_out + x in the conv's epilogue; linear attention at 8x8 / 16x16 with 64 / 128 channels: to_qkv inside the core)
    nq = sum("+qkv" in lb for lb in labels)
    n_ = hw * hw                                       # (n <= 64: one launch, ragged below 64; multiples of 256 beyond 256: the chunked form)
    assert nq == (1 if dim in (64, 128) and (n_ <= 64 or (kind == "linear" and n_ % 256 == 0)) else 0), labels
    assert sum(lb.startswith("conv1x1_ln") for lb in labels) + nq == 2, labels
    assert not any(lb.startswith("chan_layernorm") or lb.startswith("nfdp_add") for lb in labels), labels
    if nq:                                                                       # the same products, the same core: rounding level (q / k / v rows 1-2 ulp)
        K.set_attention_qkv_fused(False)
        try:
            assert_close(fused, _attn_block_nhwc(blk, x, "f32"), 1e-5, "to_qkv inside the attention core vs two launches")
        finally:
            K.set_attention_qkv_fused(True)
    K.set_conv_ln_fused(False)
    try:
        plain = _attn_block_nhwc(blk, x, "f32")
    finally:
        K.set_conv_ln_fused(True)
    assert_close(fused, plain, 2e-5, "LayerNorms inside the convs vs five launches")
    # torch reference of the block (reference gaussian_diffusion.py:200-209, :305-358)
    xc = x.permute(0, 3, 1, 2).double().cpu()
    ln = lambda t, g: (t - t.mean(1, keepdim=True)) * (t.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() * g.double().cpu()
    xn = ln(xc, blk.fn.norm.g)
    qkv = torch.nn.functional.conv2d(xn, att.to_qkv.weight.double().cpu())
    q, k, v = [t.reshape(b, att.heads, att.dim_head, hw * hw) for t in qkv.chunk(3, dim=1)]
    if kind == "linear":
        q = q.softmax(dim=-2) * att.scale
        k = k.softmax(dim=-1)
        v = v / (hw * hw)
        ctx = torch.einsum("bhdn,bhen->bhde", k, v)
        out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, -1, hw, hw)
        out = torch.nn.functional.conv2d(out, att.to_out[0].weight.double().cpu(), att.to_out[0].bias.double().cpu())
        out = ln(out, att.to_out[1].g)
    else:
        sim = torch.einsum("bhdi,bhdj->bhij", q * att.scale, k)
        out = torch.einsum("bhij,bhdj->bhid", sim.softmax(dim=-1), v)
        out = out.permute(0, 1, 3, 2).reshape(b, -1, hw, hw)
        out = torch.nn.functional.conv2d(out, att.to_out.weight.double().cpu(), att.to_out.bias.double().cpu())
    assert_close(fused.permute(0, 3, 1, 2), out + xc, 5e-5, "attention block vs torch fp64")


@pytest.mark.parametrize("b,hw,cin,c2,cout,groups", [(64, 16, 64, 0, 64, 8), (64, 16, 64, 64, 64, 8), (64, 8, 128, 0, 128, 8), (64, 8, 128, 64, 128, 8),
                                                     (3, 16, 64, 0, 128, 16), (5, 8, 64, 0, 64, 4), (2, 32, 32, 0, 64, 8), (20, 64, 64, 0, 64, 16), (18, 64, 64, 64, 64, 16),
                                                     (64, 8, 128, 128, 128, 8), (9, 16, 64, 0, 128, 2), (6, 16, 96, 32, 64, 8)])
@torch.no_grad()
def test_resnet_block_with_the_first_groupnorm_inside_the_second_conv(b, hw, cin, c2, cout, groups):
    """ResnetBlock (reference gaussian_diffusion.py:279-302) where a conv workgroup's tile is a part of an image: block1's conv leaves the
    fp64 group sums of its tiles (nfdp_conv2d_stats_nhwc_f32), block2's conv normalises its input while staging it
    (nfdp_conv2d_gnin_nhwc_f32) - against the launch-per-op form (same conv, same GroupNorm expressions; the statistics are sums of
    per-tile fp64 sums instead of a two-pass fp64 mean / variance: 2e-6) and against torch fp64.  Two-source input (the skip
    connections), 4 / 8 / 16 channels per group, 2-64 tiles per image; the 64x64 cases are the largest CelebA latent part's stage (more than
    1024 output tiles: the LDS-staged kernel keeps them in the split-K family)."""
    from diffusion_prior.gaussian_diffusion import ResnetBlock
    torch.manual_seed(b * 131 + hw + cin + c2 + cout)
    blk = ResnetBlock(cin + c2, cout, time_emb_dim=256, groups=groups).to(DEV)
    for p_ in blk.parameters():
        p_.data.copy_(torch.randn_like(p_) * (0.15 if p_.dim() > 1 else 0.3) + (1.0 if p_.dim() == 1 and p_.shape[0] == cout else 0.0))
    x = torch.randn(b, hw, hw, cin, device=DEV)
    x2 = torch.randn(b, hw, hw, c2, device=DEV) if c2 else None
    temb = torch.randn(b, 256, device=DEV)
    inp = (x, x2) if c2 else x
    # the group sums against torch, the raw conv output against the plain conv (the same kernel: bit-identical)
    cen = K.compute_enum("f32")
    raw, sums, tile = K.conv2d_stats(x, blk.block1.proj.packed(cen), cout, 3, blk.block1.proj.bias, groups, x2=x2)
    plain = blk.block1.proj.forward_nhwc(x, "f32", x2=x2) if c2 else K.conv2d(x, blk.block1.proj.packed(cen), cout, 3, add=blk.block1.proj.bias,
                                                                               compute=cen, out_dtype=K.F32)
    assert torch.equal(raw, plain)
    assert (hw * hw) % tile == 0 and sums.shape == (b * hw * hw // tile, groups, 2)
    # (the tiles of an image are flat pixel ranges or, on wide images, 8 x 8 blocks: what is pinned is an image's total)
    rt = raw.double().reshape(b, hw * hw, groups, cout // groups)
    st = sums.reshape(b, hw * hw // tile, groups, 2).sum(dim=1)
    assert_close(st[..., 0], rt.sum(dim=(1, 3)), 1e-12, "group sums")
    assert_close(st[..., 1], (rt * rt).sum(dim=(1, 3)), 1e-12, "group sums of squares")
    K.start_trace()
    carried = blk.forward_nhwc(inp, temb, "f32")
    labels = [e["label"] for e in K.stop_trace()]
    big = (b * hw * hw // 64) * (-(-cout // 64)) > K.GN_ON_LOAD_MAX_TILES       # more than a chip-wide wave of conv workgroups:
    if big:                                                                      # both norms as streaming launches that read the convs' sums
        assert sum("+sums" in lb for lb in labels) == 2 and sum("gn-in" in lb for lb in labels) == 0, labels
        assert sum(lb.startswith("groupnorm_silu_sums") for lb in labels) == 2, labels
    else:
        assert sum("+sums" in lb and "gn-in" not in lb for lb in labels) == 1 and sum("gn-in" in lb for lb in labels) == 1, labels
        assert sum(lb.startswith("groupnorm_silu") for lb in labels) == 1, labels   # block2's norm (from its conv's sums where supported)
    K.set_conv_gn_carried(False)
    try:
        K.start_trace()
        plain_out = blk.forward_nhwc(inp, temb, "f32")
        labels = [e["label"] for e in K.stop_trace()]
        assert not any("gn-in" in lb or "+sums" in lb for lb in labels), labels
    finally:
        K.set_conv_gn_carried(True)
    assert_close(carried, plain_out, 2e-6, "GroupNorm carried into the next conv vs launch per op")
    # torch fp64 reference of the block
    xc = (torch.cat((x, x2), dim=3) if c2 else x).permute(0, 3, 1, 2).double().cpu()
    F_ = torch.nn.functional

    def ws(wt):
        wd = wt.double().cpu()
        m = wd.mean(dim=(1, 2, 3), keepdim=True)
        v = wd.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
        return (wd - m) * (v + 1e-5).rsqrt()
    ssr = F_.linear(F_.silu(temb.double().cpu()), blk.mlp[1].weight.double().cpu(), blk.mlp[1].bias.double().cpu())
    sc, sh = ssr[:, :cout, None, None], ssr[:, cout:, None, None]
    h = F_.conv2d(xc, ws(blk.block1.proj.weight), blk.block1.proj.bias.double().cpu(), padding=1)
    h = F_.group_norm(h, groups, blk.block1.norm.weight.double().cpu(), blk.block1.norm.bias.double().cpu(), 1e-5)
    h = F_.silu(h * (sc + 1) + sh)
    h = F_.conv2d(h, ws(blk.block2.proj.weight), blk.block2.proj.bias.double().cpu(), padding=1)
    h = F_.silu(F_.group_norm(h, groups, blk.block2.norm.weight.double().cpu(), blk.block2.norm.bias.double().cpu(), 1e-5))
    r = F_.conv2d(xc, blk.res_conv.weight.double().cpu(), blk.res_conv.bias.double().cpu()) if cin + c2 != cout else xc
    assert_close(carried.permute(0, 3, 1, 2), h + r, 5e-5, "ResnetBlock vs torch fp64")


@pytest.mark.parametrize("b,hw,c,groups,with_res", [(64, 4, 64, 8, True), (260, 8, 64, 16, False), (130, 4, 128, 8, True)])
@torch.no_grad()
def test_conv_normalising_on_load_with_its_own_block_epilogue(b, hw, c, groups, with_res):
    """nfdp_conv2d_gnin_nhwc_f32 with out_groups > 0 (tiles that hold whole images: the input GroupNorm from one tile's sums, the conv's own
    GroupNorm -> SiLU -> + residual in its epilogue) against conv -> GroupNorm launch -> one-launch Block."""
    torch.manual_seed(b + hw + c)
    w1, w2 = torch.randn(c, c, 3, 3, device=DEV) * 0.05, torch.randn(c, c, 3, 3, device=DEV) * 0.05
    b1, b2 = torch.randn(c, device=DEV) * 0.1, torch.randn(c, device=DEV) * 0.1
    g1, be1, g2, be2 = [torch.randn(c, device=DEV) * 0.3 + o for o in (1.0, 0.0, 1.0, 0.0)]
    ss = torch.randn(b, 2 * c, device=DEV) * 0.3
    res = torch.randn(b, hw, hw, c, device=DEV) if with_res else None
    x = torch.randn(b, hw, hw, c, device=DEV)
    cen = K.compute_enum("f32")
    p1, p2 = K.conv_pack(w1, cen), K.conv_pack(w2, cen)
    assert K.lib().nfdp_conv2d_gnin_supported(b, hw, hw, c, c, groups, groups)
    raw, sums, tile = K.conv2d_stats(x, p1, c, 3, b1, groups)
    assert tile == hw * hw or (hw * hw) % tile == 0
    got = K.conv2d_gnin(raw, sums, tile, groups, g1, be1, ss, p2, c, b2, out_groups=groups, out_gamma=g2, out_beta=be2, residual=res)
    h = K.groupnorm_silu(K.conv2d(x, p1, c, 3, add=b1, compute=cen, out_dtype=K.F32), groups, g1, be1, ss, 1e-5)
    want = K.conv2d_gn_silu(h, p2, c, 3, b2, groups, g2, be2, None, 1e-5, res)
    assert_close(got, want, 2e-6, "normalise-on-load + Block epilogue vs three launches")


@pytest.mark.parametrize("b,hw,cin,cout", [(64, 8, 128, 64), (3, 4, 64, 64), (2, 16, 96, 128), (64, 2, 128, 128)])
@torch.no_grad()
def test_upsample_conv_in_one_launch_is_bit_identical(b, hw, cin, cout):
    """Upsample (reference gaussian_diffusion.py:169-173: nearest x2 -> Conv2d 3x3) with the upsampling as an index map of the conv's
    staging (nfdp_conv2d_up2_nhwc_f32) against upsample launch + conv: the same kernel on the same values - bit for bit."""
    torch.manual_seed(b + hw + cin)
    x = torch.randn(b, hw, hw, cin, device=DEV)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    bias = torch.randn(cout, device=DEV)
    packed = K.conv_pack(w, K.F32)
    assert K.conv2d_up2_supported(x, cout, bias)
    got = K.conv2d_up2(x, packed, cout, bias)
    want = K.conv2d(K.upsample2x(x), packed, cout, 3, add=bias, compute=K.compute_enum("f32"), out_dtype=K.F32)
    assert got.shape == (b, 2 * hw, 2 * hw, cout) and torch.equal(got, want)
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x.permute(0, 3, 1, 2).double().cpu(), scale_factor=2, mode="nearest"),
                                     w.double().cpu(), bias.double().cpu(), padding=1)
    assert_close(got.permute(0, 3, 1, 2), ref, 2e-5, "upsample + conv vs torch fp64")


@pytest.mark.parametrize("b,hw,c,cout", [(64, 16, 64, 64), (3, 8, 64, 128), (2, 32, 96, 64), (5, 4, 128, 128)])
@torch.no_grad()
def test_downsample_as_a_stride_two_conv(b, hw, c, cout):
    """Downsample (reference gaussian_diffusion.py:176-180: Rearrange 'b c (h p1) (w p2) -> b (c p1 p2) h w' -> Conv2d 1x1) as the 2x2 /
    stride-2 conv it is on the un-rearranged tensor (nfdp_conv2d_squeeze_nhwc_f32) against squeeze launch + 1x1 conv (2e-6: another K
    association) and against torch fp64 with einops' channel order spelled out."""
    torch.manual_seed(b + hw + c)
    x = torch.randn(b, hw, hw, c, device=DEV)
    w = torch.randn(cout, 4 * c, 1, 1, device=DEV) * 0.05
    bias = torch.randn(cout, device=DEV)
    assert K.conv2d_squeeze_supported(x, cout, bias)
    got = K.conv2d_squeeze(x, K.conv_pack(w.view(cout, c, 2, 2), K.F32), cout, bias)
    want = K.conv2d(K.squeeze(x), K.conv_pack(w, K.F32), cout, 1, add=bias, compute=K.compute_enum("f32"), out_dtype=K.F32)
    assert_close(got, want, 2e-6, "stride-2 conv vs squeeze + 1x1")
    xc = x.permute(0, 3, 1, 2).double().cpu()
    xr = xc.reshape(b, c, hw // 2, 2, hw // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, 4 * c, hw // 2, hw // 2)   # (c p1 p2)
    ref = torch.nn.functional.conv2d(xr, w.double().cpu(), bias.double().cpu())
    assert_close(got.permute(0, 3, 1, 2), ref, 2e-5, "Downsample vs torch fp64")


@pytest.mark.parametrize("b,h,w,cin,cout", [(64, 16, 16, 2, 64), (3, 8, 8, 4, 64), (2, 20, 12, 3, 20), (5, 4, 4, 1, 128), (1, 64, 64, 2, 64), (3, 32, 32, 6, 64), (2, 16, 16, 8, 128),
                                               (2, 8, 8, 5, 64), (2, 8, 8, 7, 64)])
@torch.no_grad()
def test_direct_7x7_conv_over_few_input_channels(b, h, w, cin, cout):
    """Unet.init_conv (reference gaussian_diffusion.py:383) on the direct fp32 kernel (nfdp_conv_k7_narrow_f32: 64 pixels x 64 outputs per
    workgroup, FMA chains over (tap, channel)) against the implicit-GEMM conv and torch fp64; 1-8 input channels, non-square / non-power-of-two
    images, partial pixel tiles."""
    torch.manual_seed(b + h + cin + cout)
    x = torch.randn(b, h, w, cin, device=DEV)
    wt = torch.randn(cout, cin, 7, 7, device=DEV) * 0.1
    bias = torch.randn(cout, device=DEV)
    assert K.conv_k7_narrow_supported(x, cout, bias)
    got = K.conv_k7_narrow(x, K.conv_k7_pack(wt), cout, bias)
    want = K.conv2d(x, K.conv_pack(wt, K.F32), cout, 7, add=bias, compute=K.compute_enum("f32"), out_dtype=K.F32)
    assert_close(got, want, 5e-6, "direct 7x7 vs implicit GEMM")
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), bias.double().cpu(), padding=3)
    assert_close(got.permute(0, 3, 1, 2), ref, 1e-5, "direct 7x7 vs torch fp64")


@pytest.mark.parametrize("b,hw,cin,c2,cout,groups", [(64, 4, 64, 64, 64, 8), (64, 2, 128, 64, 128, 8), (64, 4, 64, 0, 128, 8), (7, 4, 128, 128, 128, 16)])
@torch.no_grad()
def test_resnet_block_of_whole_image_tiles_with_the_res_conv_in_block1s_launch(b, hw, cin, c2, cout, groups):
    """ResnetBlock where a conv tile holds whole images (the 4x4 / 2x2 stages): block1 as one launch (conv + GroupNorm + SiLU) that also
    forms the 1x1 res_conv of the same input (nfdp_conv2d_gn_silu_res_nhwc_f32) against the form with a res_conv launch (the 1x1 products in
    another K association: 2e-6) and the launch count."""
    from diffusion_prior.gaussian_diffusion import ResnetBlock
    torch.manual_seed(b * 7 + hw + cin + c2 + cout)
    blk = ResnetBlock(cin + c2, cout, time_emb_dim=256, groups=groups).to(DEV)
    for p_ in blk.parameters():
        p_.data.copy_(torch.randn_like(p_) * (0.15 if p_.dim() > 1 else 0.3) + (1.0 if p_.dim() == 1 and p_.shape[0] == cout else 0.0))
    x = torch.randn(b, hw, hw, cin, device=DEV)
    x2 = torch.randn(b, hw, hw, c2, device=DEV) if c2 else None
    temb = torch.randn(b, 256, device=DEV)
    inp = (x, x2) if c2 else x
    blk.forward_nhwc(inp, temb, "f32")                                       # (packs the weights)
    K.start_trace()
    fused = blk.forward_nhwc(inp, temb, "f32")
    labels = [e["label"] for e in K.stop_trace()]
    assert sum("+gn+res" in lb for lb in labels) == 1 and len([lb for lb in labels if not lb.startswith("nfdp_linear")]) == 2, labels
    K.set_conv_res_fused(False)
    try:
        plain = blk.forward_nhwc(inp, temb, "f32")
    finally:
        K.set_conv_res_fused(True)
    assert_close(fused, plain, 2e-6, "res_conv in block1's launch vs its own launch")


@pytest.mark.parametrize("b,h,w,c1,c2,cout", [(20, 64, 64, 64, 0, 64), (5, 32, 32, 64, 64, 64), (3, 64, 32, 128, 0, 128), (18, 64, 64, 64, 64, 64)])
@torch.no_grad()
def test_block_tiles_of_the_staged_conv_are_bit_identical_to_flat_tiles(b, h, w, c1, c2, cout):
    """conv3x3_ldsx_kernel on wide images: a workgroup's 64 pixels as an 8 x 8 block of one image (100 staged pixels) instead of a flat range
    (64 + 2 (W + 1)) - the same K order per output, so the same bits as the flat tiles (debug switch 2); one- and two-source inputs,
    non-square images, more than 1024 output tiles."""
    torch.manual_seed(b + h + w + c1 + c2)
    x = torch.randn(b, h, w, c1, device=DEV)
    x2 = torch.randn(b, h, w, c2, device=DEV) if c2 else None
    wt = torch.randn(cout, c1 + c2, 3, 3, device=DEV) * 0.05
    bias = torch.randn(cout, device=DEV)
    packed = K.conv_pack(wt, K.F32)
    cen = K.compute_enum("f32")
    run = (lambda: K.conv2d_cat(x, x2, packed, cout, 3, add=bias)) if c2 else (lambda: K.conv2d(x, packed, cout, 3, add=bias, compute=cen, out_dtype=K.F32))
    blocks = run()
    K.lib().nfdp_debug_set_conv_ldsx(2)
    try:
        flat = run()
    finally:
        K.lib().nfdp_debug_set_conv_ldsx(1)
    assert torch.equal(blocks, flat)
    xc = (torch.cat((x, x2), dim=3) if c2 else x).permute(0, 3, 1, 2).double().cpu()
    ref = torch.nn.functional.conv2d(xc, wt.double().cpu(), bias.double().cpu(), padding=1)
    assert_close(blocks.permute(0, 3, 1, 2), ref, 2e-5, "staged conv on block tiles vs torch fp64")


@torch.no_grad()
def test_unet_in_exact_fp32_products_mode_takes_the_unfused_paths_and_agrees():
    """nfdp_hip.set_f32_products("exact") (fp32 MFMA chain instead of the bf16 splits): the split-product fusions (GroupNorm carried, res_conv,
    Upsample, to_qkv inside attention, LayerNorms in the 1x1 convs) are gated off, Downsample-as-a-conv and the direct 7x7 kernel stay; the
    U-Net's output agrees with the default mode to the products' rounding."""
    from diffusion_prior import Unet
    torch.manual_seed(3)
    u = Unet(dim=64, dim_mults=(1, 2), channels=2, resnet_block_groups=8).to(DEV)
    x = torch.randn(8, 16, 16, 2, device=DEV)
    t = torch.randint(0, 1000, (8,), device=DEV)
    want = u.forward_nhwc(x, t)
    K.set_f32_products("exact")
    try:
        K.start_trace()
        got = u.forward_nhwc(x, t)
        labels = [e["label"] for e in K.stop_trace()]
    finally:
        K.set_f32_products("bf16x6")
    assert not any(("gn-in" in lb) or ("+qkv" in lb) or lb.startswith("conv1x1_ln") or ("up2" in lb) or ("+res" in lb) for lb in labels), labels
    assert any("squeeze" in lb and lb.startswith("conv_splitk") for lb in labels) and any(lb.startswith("conv_k7_narrow") for lb in labels), labels
    assert_close(got, want, 2e-5, "U-Net: exact fp32 products vs split products")
