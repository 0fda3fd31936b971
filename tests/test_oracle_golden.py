"""Pins oracle/ (the CPU restatement) against golden vectors produced by the reference itself.

CPU-only; runs in the build container and on the GPU box alike (no /root/reference needed).
Tolerance: the oracle executes the same torch-CPU ops as the reference, so results agree to fp32
rounding noise; 2e-6 relative is asserted (most cases are bit-identical).
"""
import numpy as np
import pytest
import torch

from oracle import nf_oracle as O
from oracle import diffusion_oracle as D
from tests.util import load_golden, meta_of, synth_sd, T, assert_close, relerr
from tests.golden.synth import synth_tensor

TOL = 2e-6
torch.set_num_threads(8)


@pytest.fixture(scope="module")
def ops():
    return load_golden("ops.npz")


@pytest.mark.parametrize("c", [4, 12, 48])
def test_actnorm(ops, c):
    x = T(ops[f"actnorm{c}_x"])
    s, b = O.actnorm_init(x)
    assert_close(s, ops[f"actnorm{c}_scale"], TOL, "scale")
    assert_close(b, ops[f"actnorm{c}_bias"], TOL, "bias")
    y, ldj = O.actnorm_fwd(x, s, b, torch.zeros(4, dtype=torch.float64))
    assert_close(y, ops[f"actnorm{c}_y"], TOL, "y")
    assert_close(ldj, ops[f"actnorm{c}_ldj"], TOL, "ldj")
    assert_close(O.actnorm_inv(y, s, b), ops[f"actnorm{c}_inv"], TOL, "inv")


@pytest.mark.parametrize("c", [4, 12, 48])
def test_invconv(ops, c):
    x, w = T(ops[f"invconv{c}_x"]), T(ops[f"invconv{c}_w"])
    y, ldj = O.invconv_fwd(x, w, torch.zeros(4, dtype=torch.float64))
    assert_close(y, ops[f"invconv{c}_y"], TOL, "y")
    assert_close(ldj, ops[f"invconv{c}_ldj"], TOL, "ldj")
    assert_close(O.invconv_inv(y, w), ops[f"invconv{c}_inv"], 1e-5, "inv")


def test_squeeze_bit_exact(ops):
    x = T(ops["squeeze_x"])
    y = O.squeeze(x)
    assert torch.equal(y, T(ops["squeeze_y"]))
    assert torch.equal(O.unsqueeze(y), T(ops["squeeze_inv"]))
    assert torch.equal(O.unsqueeze(y), x)


@pytest.mark.parametrize("tag,c,nf", [("a", 4, 64), ("b", 12, 64), ("c", 4, 512), ("d", 48, 512)])
def test_coupling(ops, tag, c, nf):
    p = "flows.0.affcoupling.net."
    spec = [("0._Conv2dActNorm__conv.weight", (nf, c // 2, 3, 3)), ("0._Conv2dActNorm__actnorm.scale", (nf, 1, 1)),
            ("0._Conv2dActNorm__actnorm.bias", (nf, 1, 1)), ("2._Conv2dActNorm__conv.weight", (nf, nf, 1, 1)),
            ("2._Conv2dActNorm__actnorm.scale", (nf, 1, 1)), ("2._Conv2dActNorm__actnorm.bias", (nf, 1, 1)),
            ("4.weight", (c, nf, 3, 3)), ("4.bias", (c,)), ("4.logs", (1, c, 1, 1))]
    sd = {p + k: v for k, v in synth_sd(spec, 2, prefix=p).items()}
    x = T(ops[f"coup{tag}_x"])
    y, ldj = O.coupling_fwd(x, sd, p, torch.zeros(2, dtype=torch.float64))
    assert_close(y, ops[f"coup{tag}_y"], TOL, "y")
    assert_close(ldj, ops[f"coup{tag}_ldj"], TOL, "ldj")
    assert_close(O.coupling_inv(y, sd, p), ops[f"coup{tag}_inv"], TOL, "inv")


@pytest.mark.parametrize("c", [4, 24])
def test_split(ops, c):
    p = "blocks.0.split."
    spec = [("conv.weight", (c, c // 2, 3, 3)), ("conv.bias", (c,)), ("conv.logs", (1, c, 1, 1))]
    sd = {p + k: v for k, v in synth_sd(spec, 3, prefix=p).items()}
    x = T(ops[f"split{c}_x"])
    y, z, logp = O.split_fwd(x, sd, p, torch.zeros(2, dtype=torch.float64))
    assert torch.equal(y, T(ops[f"split{c}_y"])) and torch.equal(z, T(ops[f"split{c}_z"]))
    assert_close(logp, ops[f"split{c}_logp"], TOL, "logp")
    assert torch.equal(O.split_inv(y, sd, p, z, 1.0, None), T(ops[f"split{c}_inv_given"]))
    inv = O.split_inv(y, sd, p, None, 0.7, T(ops[f"split{c}_eps"]))
    assert_close(inv, ops[f"split{c}_inv_sampled"], TOL, "sampled")


@pytest.mark.parametrize("c", [16, 48])
def test_gaussian_prior(ops, c):
    spec = [("_GaussianPrior__conv.weight", (2 * c, 2 * c, 3, 3)), ("_GaussianPrior__conv.bias", (2 * c,)),
            ("_GaussianPrior__conv.logs", (1, 2 * c, 1, 1))]
    sd = synth_sd(spec, 4)
    x = T(ops[f"gprior{c}_x"])
    assert_close(O.gaussian_prior_logp(x, sd), ops[f"gprior{c}_logp"], TOL, "logp")
    s = O.gaussian_prior_sample((3, c, 4, 4), sd, 0.8, T(ops[f"gprior{c}_eps"]))
    assert_close(s, ops[f"gprior{c}_sample"], TOL, "sample")


def test_pre_post_bpd(ops):
    b01 = T(ops["pp_u8"]).float() / 255.0
    for nb in (5, 8):
        assert torch.equal(O.preprocess(b01, nb), T(ops[f"pre{nb}"]))
    assert torch.equal(O.postprocess(T(ops["post_x"]), 32.0), T(ops["post5"]))
    assert_close(O.bits_per_dim(T(ops["bpd_ll"]), 32.0, 3072.0), ops["bpd"], 1e-12, "bpd")
    assert np.array_equal(np.array(O.output_shapes(3, 3, 32)), ops["shapes_3_3_32"])
    assert np.array_equal(np.array(O.output_shapes(5, 3, 128)), ops["shapes_5_3_128"])
    with pytest.raises(ValueError):
        O.output_shapes(3, 3, 30 // 2 * 2 + 1)


@pytest.mark.parametrize("tag", ["M", "N", "D", "C"])
def test_glow_composite(tag):
    """M / N / D: 32x32 inputs; C: BASELINE configs[4]'s image size and depth (3x128x128, L = 5; K reduced to 2)."""
    g = load_golden(f"glow_{tag}.npz")
    m = meta_of(g)
    L, K, B, C = m["L"], m["K"], m["B"], m["C"]
    sd = synth_sd(m["flow_spec"], m["seed"], profile=m.get("profile"))
    psd = synth_sd(m["prior_spec"], m["seed"], profile=m.get("profile"))
    x = T(g["x"])
    assert x.shape[-1] == m.get("size", 32)
    zs, ldj, logp, bpd = O.fwd_nll(x, sd, psd, L, K)
    for i, z in enumerate(zs):
        assert_close(z, g[f"z{i}"], TOL, f"z{i}")
    assert_close(ldj, g["ldj"], TOL, "ldj")
    assert_close(logp, g["logp"], TOL, "logp")
    assert_close(bpd, g["bpd"], TOL, "bpd")
    assert_close(O.glow_inv(zs, sd, L, K), g["x_rec"], 2e-5, "x_rec")
    eps = [T(g[f"eps{j}"]) for j in range(L - 1)]
    assert_close(O.glow_inv([zs[-1]], sd, L, K, 0.7, eps), g["x_sampled"], 2e-5, "x_sampled")
    xs0 = O.glow_inv([zs[-1]], sd, L, K, 0.0, [torch.zeros_like(e) for e in eps])
    post = O.postprocess(xs0, 32.0)
    assert (post.int() - T(g["x_sampled_post"]).int()).abs().max() <= 8   # one 5-bit bin at most (floor edge)
    assert (post != T(g["x_sampled_post"])).float().mean() < 1e-3


def test_glow_data_dependent_init():
    g = load_golden("glow_init.npz")
    m = meta_of(g)
    L, K, B, C = m["L"], m["K"], m["B"], m["C"]
    shapes = {}
    # init-state dict: zeros everywhere except synthesised conv / invconv weights (as make_goldens.gen_glow_init)
    from tests.util import load_golden as _lg
    spec = meta_of(_lg("glow_N.npz"))["flow_spec"]
    sd = {}
    for k, s in spec:
        if k.endswith("_Conv2dActNorm__conv.weight") or k.endswith("invconv2d.weight"):
            sd[k] = T(synth_tensor(k, s, m["seed"]))
        elif k.endswith("is_initialized"):
            sd[k] = torch.tensor(0, dtype=torch.uint8)
        else:
            sd[k] = torch.zeros(*s)
    c_last = 2 ** (L + 1) * C
    psd = {"_GaussianPrior__conv.weight": torch.zeros(2 * c_last, 2 * c_last, 3, 3),
           "_GaussianPrior__conv.bias": torch.zeros(2 * c_last), "_GaussianPrior__conv.logs": torch.zeros(1, 2 * c_last, 1, 1)}
    zs, ldj, logp, bpd = O.fwd_nll(T(g["x"]), sd, psd, L, K, init=True)
    for i, z in enumerate(zs):
        assert_close(z, g[f"z{i}"], 1e-5, f"z{i}")
    assert_close(ldj, g["ldj"], 1e-6, "ldj")
    assert_close(bpd, g["bpd"], 1e-6, "bpd")
    for k in [k for k in g if k.startswith("p:")]:
        assert_close(sd[k[2:]], g[k], 1e-5, k)
    zs2, ldj2, logp2 = O.glow_fwd(T(g["x2"]), sd, L, K, torch.zeros(B, dtype=torch.float64), torch.zeros(B, dtype=torch.float64))
    assert_close(zs2[-1], g["z2_last"], 1e-5, "z2_last")
    assert_close(ldj2, g["ldj2"], 1e-6, "ldj2")


# ----------------------------------------------------------------------------- diffusion
@pytest.fixture(scope="module")
def dg():
    return load_golden("diffusion.npz")


@pytest.mark.parametrize("sched", ["cosine", "linear", "sigmoid"])
def test_schedule_buffers(dg, sched):
    buf = D.diffusion_buffers(sched, 1000)
    for name in D.BUFFER_NAMES:
        assert_close(buf[name], dg[f"buf_{sched}_{name}"], 1e-6, name)
    with pytest.raises(ValueError):
        D.betas_for("nope", 10)


@pytest.mark.parametrize("tag", ["m0", "m2", "c0", "c5"])
def test_unet_forward(dg, tag):
    m = meta_of(dg)[tag]
    sd = synth_sd(m["spec"], 21)
    y = D.unet_forward(T(dg[f"unet{tag}_x"]), T(dg[f"unet{tag}_t"]), sd, 64, m["mults"], m["groups"])
    assert_close(y, dg[f"unet{tag}_y"], 1e-5, "unet")


def test_unet_celeba_part():
    """The U-Net of BASELINE configs[4]'s first latent part (6x64x64, dim_mults [1,2,2], groups 16): forward, p_losses and one
    p_sample against the reference's outputs (tools/make_goldens.py celeba)."""
    cg = load_golden("unet_celeba.npz")
    m = meta_of(cg)
    sd = synth_sd(m["spec"], 27)
    model = lambda x, t: D.unet_forward(x, t, sd, 64, m["mults"], m["groups"])
    x, t, eps = T(cg["x"]), T(cg["t"]), T(cg["eps"])
    assert_close(model(x, t), cg["y"], 1e-5, "unet 6x64x64")
    buf = D.diffusion_buffers("cosine", 1000)
    assert_close(D.p_losses(buf, model, x * 0.8, t, eps, "l2"), cg["p_losses"], 1e-5, "p_losses")
    img, xs = D.p_sample(buf, model, x * 0.8, int(t[0]), T(cg["psample_noise"]))
    assert_close(img, cg["psample_img"], 1e-5, "p_sample img")
    assert_close(xs, cg["psample_x0"], 1e-5, "p_sample x0")


def test_diffusion_process(dg):
    m = meta_of(dg)["m0"]
    sd = synth_sd(m["spec"], 21)
    model = lambda x, t: D.unet_forward(x, t, sd, 64, m["mults"], m["groups"])
    buf = D.diffusion_buffers("cosine", 1000)
    x0, eps, t = T(dg["d_x0"]), T(dg["d_eps"]), T(dg["d_t"])
    assert_close(D.q_sample(buf, x0, t, eps), dg["d_q_sample"], TOL, "q_sample")
    assert_close(D.predict_start_from_noise(buf, x0, t, eps), dg["d_pred_x0"], TOL, "pred_x0")
    pm, pv, plv = D.q_posterior(buf, x0, eps, t)
    assert_close(pm, dg["d_post_mean"], TOL, "pm")
    assert_close(pv, dg["d_post_var"], TOL, "pv")
    assert_close(plv, dg["d_post_logvar"], TOL, "plv")
    assert_close(D.p_losses(buf, model, x0, t, eps, "l2"), dg["d_p_losses_l2"], 1e-5, "l2")
    assert_close(D.p_losses(buf, model, x0, t, eps, "l1"), dg["d_p_losses_l1"], 1e-5, "l1")
    assert_close(D.p_losses(buf, model, x0, T(dg["d_fwd_t"]), T(dg["d_fwd_noise"]), "l2"), dg["d_fwd_loss"], 1e-5, "fwd")
    for step in (0, 1, 500, 999):
        img, xs = D.p_sample(buf, model, x0, step, T(dg[f"d_psample{step}_noise"]))
        assert_close(img, dg[f"d_psample{step}_img"], 1e-5, f"img{step}")
        assert_close(xs, dg[f"d_psample{step}_x0"], 1e-5, f"x0{step}")
    with pytest.raises(ValueError):
        D.p_losses(buf, model, x0, t, eps, "huber")


def test_sampling_loops(dg):
    m = meta_of(dg)["m0"]
    sd = synth_sd(m["spec"], 21)
    model = lambda x, t: D.unet_forward(x, t, sd, 64, m["mults"], m["groups"])
    buf8 = D.diffusion_buffers("cosine", 8)
    out = D.p_sample_loop(buf8, model, T(dg["loop8_xT"]), list(T(dg["loop8_noises"])))
    assert_close(out, dg["loop8_out"], 2e-5, "loop8")
    out = D.ddim_sample(buf8, model, T(dg["ddim_xT"]), 4, 1.0, list(T(dg["ddim_noises"])))
    assert_close(out, dg["ddim_out"], 2e-5, "ddim")


# ----------------------------------------------------------------------------- objectives / self-conditioning (variants)
@pytest.fixture(scope="module")
def xg():
    return load_golden("diffusion_variants.npz")


@pytest.mark.parametrize("obj", ["pred_x0", "pred_v"])
def test_objective_variants(xg, obj):
    """The oracle's restatement of model_predictions / p_sample / the loops / p_losses for pred_x0 and pred_v against outputs of the
    reference itself (tools/make_goldens.py variants)."""
    sd = synth_sd(meta_of(xg)["unet"], 21)
    model = lambda x, t, sc=None: D.unet_forward(x, t, sd, 64, (1, 2), 8)
    buf = D.diffusion_buffers("cosine", 1000)
    x0, eps, t = T(xg["x0"]), T(xg["eps"]), T(xg["t"])
    assert_close(D.p_losses_v(buf, model, x0, t, eps, obj), xg[f"{obj}_p_losses"], 1e-5, "p_losses")
    for clip in (False, True):
        pn, xs = D.model_predictions(buf, model, x0, t, obj, clip_x_start=clip)
        assert_close(pn, xg[f"{obj}_mp_noise_clip{int(clip)}"], 2e-5, "pred_noise")
        assert_close(xs, xg[f"{obj}_mp_x0_clip{int(clip)}"], 2e-5, "x_start")
    for step in (0, 500):
        img, xs = D.p_sample_v(buf, model, x0, step, T(xg[f"{obj}_psample{step}_noise"]), obj)
        assert_close(img, xg[f"{obj}_psample{step}_img"], 2e-5, f"img{step}")
    assert_close(D.predict_v(buf, x0, t, eps), xg[f"{obj}_predict_v"], TOL, "predict_v")
    assert_close(D.predict_noise_from_start(buf, x0, t, eps), xg[f"{obj}_noise_from_start"], TOL, "noise_from_start")
    buf8 = D.diffusion_buffers("cosine", 8)
    out = D.p_sample_loop_v(buf8, model, T(xg[f"{obj}_loop8_xT"]), list(T(xg[f"{obj}_loop8_noises"])), obj)
    assert_close(out, xg[f"{obj}_loop8_out"], 5e-5, "loop8")
    out = D.ddim_sample_v(buf8, model, T(xg[f"{obj}_ddim_xT"]), 4, 1.0, list(T(xg[f"{obj}_ddim_noises"])), obj)
    assert_close(out, xg[f"{obj}_ddim_out"], 5e-5, "ddim")


def test_self_conditioning_variant(xg):
    sd = synth_sd(meta_of(xg)["unet_sc"], 24)
    model = lambda x, t, sc=None: D.unet_forward(x, t, sd, 64, (1, 2), 8, self_condition=True, x_self_cond=sc)
    buf = D.diffusion_buffers("cosine", 1000)
    x0, eps, t, sc = T(xg["x0"]), T(xg["eps"]), T(xg["t"]), T(xg["sc"])
    assert_close(model(x0, t, sc), xg["sc_unet_given"], 1e-5, "unet with self cond")
    assert_close(model(x0, t), xg["sc_unet_none"], 1e-5, "unet, zero default")
    for name, draw in (("taken", 0.1), ("skipped", 0.9)):
        got = D.p_losses_v(buf, model, x0, t, eps, self_condition=True, random_draw=draw)
        assert_close(got, xg[f"sc_p_losses_{name}"], 1e-5, f"p_losses {name}")
    img, xs = D.p_sample_v(buf, model, x0, 500, T(xg["sc_psample500_noise"]), x_self_cond=sc)
    assert_close(img, xg["sc_psample500_img"], 2e-5, "p_sample with self cond")
    buf8 = D.diffusion_buffers("cosine", 8)
    out = D.p_sample_loop_v(buf8, model, T(xg["sc_loop8_xT"]), list(T(xg["sc_loop8_noises"])), self_condition=True)
    assert_close(out, xg["sc_loop8_out"], 1e-4, "loop8 with self cond")
    out = D.ddim_sample_v(buf8, model, T(xg["sc_ddim_xT"]), 4, 1.0, list(T(xg["sc_ddim_noises"])), self_condition=True)
    assert_close(out, xg["sc_ddim_out"], 1e-3, "ddim with self cond (a chain that amplifies fp32 rounding: sqrt(1/ac_7) = 163)")


# ----------------------------------------------------------------------------- variational bound + CatFormater index maps
@pytest.fixture(scope="module")
def vg():
    return load_golden("vlb.npz")


def test_vlb_terms_and_loop(vg):
    m = meta_of(vg)
    sd = synth_sd(m["spec"], 21)
    model = lambda x, t: D.unet_forward(x, t, sd, 64, m["mults"], m["groups"])
    buf = D.diffusion_buffers("cosine", 1000)
    x0 = T(vg["x0"])
    for tag in ("a", "b", "c"):
        out = D.vb_term(buf, model, x0, T(vg[f"term_{tag}_xt"]), T(vg[f"term_{tag}_t"]))
        assert_close(out, vg[f"term_{tag}_out"], 2e-5, f"vb term {tag}")
    # the prior term is a cancellation residue (~1e-9 of O(1) operands): absolute tolerance
    assert float((D.prior_bpd(buf, x0) - T(vg["prior_T1000"])).abs().max()) <= 1e-7
    buf8 = D.diffusion_buffers("cosine", 8)
    assert_close(D.prior_bpd(buf8, x0), vg["prior_T8"], 1e-5, "prior T8")
    total = D.calc_neg_log_likelihood_loop(buf8, model, x0, list(T(vg["loop8_noises"])))
    assert_close(total, vg["loop8_total"], 2e-5, "vlb loop8")


def cat_reference_map(latents, target):
    """CatFormater.process_latents restated with the oracle's squeeze/unsqueeze (latent_formaters.py:160-184)."""
    parts = []
    for i, z in enumerate(latents):
        for _ in range(abs(target - i)):
            z = O.squeeze(z) if target - i > 0 else O.unsqueeze(z)
        parts.append(z)
    return torch.cat(parts, dim=1)


@pytest.mark.parametrize("tag", ["L3", "L5", "L4"])
def test_cat_formater_index_map(vg, tag):
    c = meta_of(vg)[f"cat_{tag}"]
    dims = O.output_shapes(c["L"], c["C"], c["size"])
    lat = [torch.arange(c["B"] * int(np.prod(d)), dtype=torch.float32).reshape(c["B"], *d) + 1000.0 * i
           for i, d in enumerate(dims)]
    out = cat_reference_map(lat, (c["L"] - 1) // 2)
    assert torch.equal(out, T(vg[f"cat_{tag}_out"]))


# ----------------------------------------------------------------------------- training-step goldens (reference autograd)
def summary64(v):
    f = torch.as_tensor(v).detach().double().reshape(-1).cpu()
    return torch.cat([f[:64], torch.stack([f.sum(), (f * f).sum(), f.abs().max()])])


def close_summary(got, ref, tol, what):
    """[first 64 | sum | sumsq | max-abs] fixtures: values relative to the tensor's max-abs, the sum relative to sqrt(N)
    times it (round-off of a long sum), sumsq and max-abs relative to themselves."""
    got, ref = summary64(got), torch.as_tensor(ref).double()
    n, scale = len(got) - 3, float(ref[-1]) if float(ref[-1]) > 0 else 1.0
    assert float((got[:n] - ref[:n]).abs().max()) <= tol * scale, f"{what}: head values"
    assert abs(float(got[-1] - ref[-1])) <= tol * scale, f"{what}: max-abs"
    assert abs(float(got[-2] - ref[-2])) <= 2 * tol * max(float(ref[-2]), 1e-30), f"{what}: sum of squares"


def test_oracle_autograd_matches_reference_training_gradients():
    """trainer.py:152-164: autograd through the oracle reproduces the reference's own parameter gradients (fixture
    tests/golden/train_N.npz, generated by running the reference), which pins the oracle as the gradient checker of
    tests/test_gpu_backward.py."""
    import json
    g = load_golden("train_N.npz")
    m = meta_of(g)
    fsd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in synth_sd(m["flow_spec"], m["seed"]).items()}
    psd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in synth_sd(m["prior_spec"], m["seed"]).items()}
    x = T(load_golden("glow_N.npz")["x"])
    _, _, _, bpd = O.fwd_nll(x, fsd, psd, m["L"], m["K"])
    assert_close(bpd, g["bpd"], 1e-6, "bpd")
    bpd.backward()
    for i, k in enumerate(json.loads(str(g["keys"]))):
        close_summary(fsd[k].grad, g[f"g{i}"], 2e-5, f"grad {k}")
    for i, k in enumerate(json.loads(str(g["pkeys"]))):
        ref = T(g[f"pg{i}"])
        if float(ref[-1]) == 0.0:
            assert psd[k].grad is None or float(psd[k].grad.abs().max()) == 0.0
        else:
            close_summary(psd[k].grad, ref, 2e-5, f"prior grad {k}")


# ----------------------------------------------------------------------------- random / learned sinusoidal time embeddings
@pytest.mark.parametrize("tag", ["learned", "random"])
def test_unet_with_learned_or_random_time_embedding(tag):
    """The oracle's U-Net with RandomOrLearnedSinusoidalPosEmb (gaussian_diffusion.py:240-255) against the reference's outputs
    (tools/make_goldens.py embeddings)."""
    eg = load_golden("unet_embeddings.npz")
    sd = synth_sd(meta_of(eg)[tag], 26)
    out = D.unet_forward(T(eg["x0"]), T(eg["t"]), sd, 64, (1, 2), 8)
    assert_close(out, eg[f"{tag}_out"], 2e-5, f"unet with {tag} embedding")
