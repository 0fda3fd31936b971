#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference).

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py
The reference never travels to the GPU box; only these small input/output fixtures do.
Weights are not stored: they are synthesised from (key, shape, seed) by tests/golden/synth.py and
loaded into the reference modules with load_state_dict(strict=True), which also pins state-dict
key compatibility.  All randomness the reference draws (dequantisation noise aside, which is an
input) is recorded by replaying torch.manual_seed on the CPU generator.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

from ref_import import import_reference
from tests.golden.synth import synth_tensor, synth_input, images_5bit, GOLDEN, DEEP

R = import_reference()
T = R["transforms"]
OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def load_synth(module, seed, overrides=None, profile=GOLDEN):
    sd = module.state_dict()
    new = {k: torch.from_numpy(np.array(synth_tensor(k, v.shape, seed, profile))) for k, v in sd.items()}
    if overrides:
        new.update(overrides)
    module.load_state_dict(new, strict=True)
    return [(k, list(v.shape)) for k, v in sd.items()]


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def n(x):
    return x.detach().cpu().numpy()


def f64z(b):
    return torch.zeros(b, dtype=torch.float64)


# ------------------------------------------------------------------ per-op goldens
@torch.no_grad()
def gen_ops():
    g = {}
    # ActNorm: data-dependent init + fwd + inv  (transforms.py:56-94)
    for c in (4, 12, 48):
        x = t(synth_input(f"actnorm{c}", (4, c, 8, 8), 1)) * 1.7 + 0.3
        m = T.ActNorm(c)
        y, ldj, _ = m.transform(x, f64z(4), None)
        g[f"actnorm{c}_x"], g[f"actnorm{c}_y"], g[f"actnorm{c}_ldj"] = n(x), n(y), n(ldj)
        g[f"actnorm{c}_scale"], g[f"actnorm{c}_bias"] = n(m.scale), n(m.bias)
        g[f"actnorm{c}_inv"] = n(m.invert(y))
    # InvConv2d  (transforms.py:116-145)
    for c in (4, 12, 48):
        x = t(synth_input(f"invconv{c}", (4, c, 8, 8), 1))
        m = T.InvConv2d(c)
        w = t(synth_tensor("x.invconv2d.weight", (c, c, 1, 1), 1))
        m.load_state_dict({"weight": w})
        y, ldj, _ = m.transform(x, f64z(4), None)
        g[f"invconv{c}_x"], g[f"invconv{c}_w"], g[f"invconv{c}_y"], g[f"invconv{c}_ldj"] = n(x), n(w), n(y), n(ldj)
        g[f"invconv{c}_inv"] = n(m.invert(y))
    # Squeeze on exact integers  (transforms.py:212-239)
    x = torch.arange(2 * 3 * 8 * 8, dtype=torch.float32).reshape(2, 3, 8, 8)
    sq = T.Squeeze()
    y = sq.transform(x, None, None)[0]
    g["squeeze_x"], g["squeeze_y"] = n(x), n(y)
    g["squeeze_inv"] = n(sq.invert(y))
    # AffineCoupling, non-zero ZeroConv  (transforms.py:166-201)
    for tag, c, nf, hw in (("a", 4, 64, 8), ("b", 12, 64, 8), ("c", 4, 512, 8), ("d", 48, 512, 4)):
        x = t(synth_input(f"coup{tag}", (2, c, hw, hw), 1))
        m = T.AffineCoupling(c, n_features=nf)
        pref = "flows.0.affcoupling."
        sd = {k: t(synth_tensor(pref + k, v.shape, 2)) for k, v in m.state_dict().items()}
        m.load_state_dict(sd, strict=True)
        y, ldj, _ = m.transform(x, f64z(2), None)
        g[f"coup{tag}_x"], g[f"coup{tag}_y"], g[f"coup{tag}_ldj"] = n(x), n(y), n(ldj)
        g[f"coup{tag}_inv"] = n(m.invert(y))
    # Split: logp, inverse with given z, inverse with sampling (eps replayed)  (transforms.py:270-309)
    for c in (4, 24):
        x = t(synth_input(f"split{c}", (2, c, 8, 8), 1))
        m = T.Split(c)
        sd = {k: t(synth_tensor("blocks.0.split." + k, v.shape, 3)) for k, v in m.state_dict().items()}
        m.load_state_dict(sd, strict=True)
        y, _, z, logp = m.transform(x, None, f64z(2))
        g[f"split{c}_x"], g[f"split{c}_y"], g[f"split{c}_z"], g[f"split{c}_logp"] = n(x), n(y), n(z), n(logp)
        g[f"split{c}_inv_given"] = n(m.invert(y, z))
        torch.manual_seed(77)
        inv = m.invert(y, None, temperature=0.7)
        torch.manual_seed(77)
        eps = torch.zeros(2, c // 2, 8, 8).normal_()
        g[f"split{c}_inv_sampled"], g[f"split{c}_eps"] = n(inv), n(eps)
    # GaussianPrior  (prior.py:53-99)
    for c in (16, 48):
        x = t(synth_input(f"gprior{c}", (3, c, 4, 4), 1))
        m = R["prior"].GaussianPrior(c)
        load_synth(m, 4)
        g[f"gprior{c}_x"], g[f"gprior{c}_logp"] = n(x), n(m.compute_log_prob(x))
        torch.manual_seed(5)
        s = m.sample((3, c, 4, 4), temperature=0.8)
        torch.manual_seed(5)
        eps = torch.zeros(3, c, 4, 4).normal_()
        g[f"gprior{c}_sample"], g[f"gprior{c}_eps"] = n(s), n(eps)
    # pre/post-process + bpd  (utils.py:175-210, 244-256)
    U = R["nf_utils"]
    u8 = t(synth_input("pp", (2, 3, 8, 8), 1, "u8"))
    b01 = u8.float() / 255.0
    for nb in (5, 8):
        g[f"pre{nb}"] = n(U.preprocess_batch(b01, nb, 2.0 ** nb))
    g["pp_u8"] = n(u8)
    xs = t(synth_input("post", (2, 3, 8, 8), 1)) * 0.4
    g["post_x"], g["post5"] = n(xs), n(U.postprocess_batch(xs, 32.0))
    ll = t(synth_input("ll", (6,), 1)).double() * 100 - 7000
    g["bpd_ll"], g["bpd"] = n(ll), n(U.calculate_loss(ll, 32.0, 32 * 32 * 3.0))
    g["shapes_3_3_32"] = np.array(U.calculate_output_shapes(3, 3, 32))
    g["shapes_5_3_128"] = np.array(U.calculate_output_shapes(5, 3, 128))
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **g)
    print("ops.npz", len(g), "arrays")


# ------------------------------------------------------------------ Glow composite goldens
@torch.no_grad()
def gen_glow(tag, C, L, K, B, seed, size=32, profile="GOLDEN"):
    """profile "DEEP" (tests/golden/synth.py): near-orthogonal 1x1 convs and gentler coupling scales, so that the SAMPLING inverse of a
    deep / large model stays well conditioned (with the GOLDEN profile the 3x128x128 L5 sample reaches |x| ~ 1e10 and the
    reference's own fp32 result is 1.4e-4 away from an fp64 evaluation - no fp32 implementation can be held to 1e-4 there)."""
    U = R["nf_utils"]
    flow = R["glow"].Glow(in_channel=C, L=L, K=K)
    prior = R["prior"].GaussianPrior(2 ** (L + 1) * C)
    prof = {"GOLDEN": GOLDEN, "DEEP": DEEP}[profile]
    spec = load_synth(flow, seed, profile=prof)
    pspec = load_synth(prior, seed, profile=prof)
    x_np, u8, dq = images_5bit(f"glow{tag}", (B, C, size, size), seed)
    x = t(x_np)
    n_pixel = size * size * 3.0                                         # trainer.py:133-134
    # sanity: same as reference preprocessing  (trainer.py:152-155)
    ref_x = U.preprocess_batch(t(u8).float() / 255.0, 5, 32.0) + t(dq) / 32.0
    assert torch.equal(ref_x, x), (ref_x - x).abs().max()
    ldj, logp = f64z(B), f64z(B)
    zs, ldj, logp = flow.transform(x, ldj, logp)
    logp_flow = logp.clone()
    logp = logp + prior.compute_log_prob(zs[-1])
    bpd = U.calculate_loss(ldj + logp, 32.0, n_pixel)
    g = {"x": x_np, "ldj": n(ldj), "logp_flow": n(logp_flow), "logp": n(logp), "bpd": n(bpd)}
    for i, z in enumerate(zs):
        g[f"z{i}"] = n(z)
    # NFBackbone path: logp=None  (normalizing_flow/__init__.py:81)
    zs2, ldj2, _ = flow.transform(x, f64z(B), None)
    assert all(torch.equal(a, b) for a, b in zip(zs, zs2))
    g["x_rec"] = n(flow.invert(zs))
    torch.manual_seed(11)
    xs = flow.invert([zs[-1]], temperature=0.7)
    torch.manual_seed(11)
    shapes = U.calculate_output_shapes(L, C, size)
    eps = [torch.zeros(B, *shapes[i]).normal_() for i in reversed(range(L - 1))]
    g["x_sampled"] = n(xs)
    for j, e in enumerate(eps):
        g[f"eps{j}"] = n(e)
    g["x_sampled_post"] = n(flow.sample([zs[-1]], postprocess_func=lambda v: U.postprocess_batch(v, 32.0), temperature=0.0))
    meta = {"C": C, "L": L, "K": K, "B": B, "seed": seed, "flow_spec": spec, "prior_spec": pspec}
    if size != 32:
        meta["size"] = size
    if profile != "GOLDEN":
        meta["profile"] = profile
    g["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, f"glow_{tag}.npz"), **g)
    print(f"glow_{tag}.npz bpd={float(bpd):.6f}  |z_last|max={float(zs[-1].abs().max()):.3f} "
          f"rec_err={float((flow.invert(zs) - x).abs().max()):.2e} |x_sampled|max={float(xs.abs().max()):.3f}")


@torch.no_grad()
def gen_glow_init(C=3, L=3, K=2, B=8, seed=5):
    """Init-state weights: reference ctor defaults (zero ZeroConvs, is_initialized=0) with synthesised
    conv / invconv weights; the first transform performs the data-dependent ActNorm init."""
    U = R["nf_utils"]
    flow = R["glow"].Glow(in_channel=C, L=L, K=K)
    sd = flow.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith("_Conv2dActNorm__conv.weight") or k.endswith("invconv2d.weight"):
            new[k] = t(synth_tensor(k, v.shape, seed))
        else:
            new[k] = v.clone()
    flow.load_state_dict(new, strict=True)
    prior = R["prior"].GaussianPrior(2 ** (L + 1) * C)
    x_np, _, _ = images_5bit("glowinit", (B, C, 32, 32), seed)
    x = t(x_np)
    zs, ldj, logp = flow.transform(x, f64z(B), f64z(B))
    logp = logp + prior.compute_log_prob(zs[-1])
    bpd = U.calculate_loss(ldj + logp, 32.0, 32 * 32 * 3.0)
    g = {"x": x_np, "ldj": n(ldj), "logp": n(logp), "bpd": n(bpd)}
    for i, z in enumerate(zs):
        g[f"z{i}"] = n(z)
    after = flow.state_dict()
    for k in ("blocks.0.flows.0.actnorm.scale", "blocks.0.flows.0.actnorm.bias",
              "blocks.1.flows.1.affcoupling.net.0._Conv2dActNorm__actnorm.scale",
              "blocks.1.flows.1.affcoupling.net.2._Conv2dActNorm__actnorm.bias",
              "final_flows.1.actnorm.scale", "final_flows.1.affcoupling.net.2._Conv2dActNorm__actnorm.scale"):
        g["p:" + k] = n(after[k])
    assert all(int(v) == 1 for k, v in after.items() if k.endswith("is_initialized"))
    # second batch through the now-initialised flow
    x2_np, _, _ = images_5bit("glowinit2", (B, C, 32, 32), seed)
    zs2, ldj2, logp2 = flow.transform(t(x2_np), f64z(B), f64z(B))
    g["x2"], g["ldj2"], g["logp2_flow"], g["z2_last"] = x2_np, n(ldj2), n(logp2), n(zs2[-1])
    g["meta"] = np.array(json.dumps({"C": C, "L": L, "K": K, "B": B, "seed": seed}))
    np.savez_compressed(os.path.join(OUT, "glow_init.npz"), **g)
    print(f"glow_init.npz bpd={float(bpd):.6f}")


# ------------------------------------------------------------------ diffusion goldens
@torch.no_grad()
def gen_diffusion():
    GD = R["gd"]
    g = {}
    # schedules + buffers  (gaussian_diffusion.py:501-634)
    dummy = GD.Unet(dim=8, dim_mults=(1,), channels=2)
    for sched in ("cosine", "linear", "sigmoid"):
        d = GD.GaussianDiffusion(dummy, image_size=4, timesteps=1000, beta_schedule=sched, auto_normalize=False)
        for name, buf in d.named_buffers(recurse=False):
            g[f"buf_{sched}_{name}"] = n(buf)
    # U-Net forwards  (gaussian_diffusion.py:451-490)
    specs = {}
    for tag, ch, size, mults, groups, B in (("m0", 2, 16, (1, 2), 8, 2), ("m2", 16, 4, (1, 2), 8, 2),
                                            ("c0", 6, 16, (1, 2), 8, 2), ("c5", 12, 16, (1, 2, 2), 16, 1)):
        u = GD.Unet(dim=64, dim_mults=mults, channels=ch, resnet_block_groups=groups)
        specs[tag] = {"spec": load_synth(u, 21), "ch": ch, "size": size, "mults": list(mults), "groups": groups}
        x = t(synth_input(f"unet{tag}", (B, ch, size, size), 21))
        tt = torch.tensor([3, 977][:B], dtype=torch.long)
        g[f"unet{tag}_x"], g[f"unet{tag}_t"], g[f"unet{tag}_y"] = n(x), n(tt), n(u(x, tt))
        if tag == "m0":
            unet_m0 = u
    # diffusion process around the m0 U-Net, T=1000 cosine  (gaussian_diffusion.py:641-715, 806-859)
    d = GD.GaussianDiffusion(unet_m0, image_size=16, timesteps=1000, beta_schedule="cosine", loss_type="l2",
                             auto_normalize=False)
    x0 = t(synth_input("x0", (2, 2, 16, 16), 22)) * 0.8
    eps = t(synth_input("eps", (2, 2, 16, 16), 23))
    tt = torch.tensor([10, 700], dtype=torch.long)
    g["d_x0"], g["d_eps"], g["d_t"] = n(x0), n(eps), n(tt)
    g["d_q_sample"] = n(d.q_sample(x0, tt, eps))
    g["d_pred_x0"] = n(d.predict_start_from_noise(x0, tt, eps))
    pm, pv, plv = d.q_posterior(x0, eps, tt)
    g["d_post_mean"], g["d_post_var"], g["d_post_logvar"] = n(pm), n(pv), n(plv)
    g["d_p_losses_l2"] = n(d.p_losses(x0, tt, noise=eps))
    d.loss_type = "l1"
    g["d_p_losses_l1"] = n(d.p_losses(x0, tt, noise=eps))
    d.loss_type = "l2"
    for step in (0, 1, 500, 999):
        torch.manual_seed(100 + step)
        img, xs = d.p_sample(x0.clone(), step)
        torch.manual_seed(100 + step)
        nz = torch.randn_like(x0)
        g[f"d_psample{step}_img"], g[f"d_psample{step}_x0"], g[f"d_psample{step}_noise"] = n(img), n(xs), n(nz)
    # forward(): t drawn by randint  (gaussian_diffusion.py:861-867)
    torch.manual_seed(9)
    loss = d(x0)
    torch.manual_seed(9)
    t_drawn = torch.randint(0, 1000, (2,)).long()
    nz = torch.randn_like(x0)
    g["d_fwd_loss"], g["d_fwd_t"], g["d_fwd_noise"] = n(loss), n(t_drawn), n(nz)
    # short full loops, T=8  (gaussian_diffusion.py:717-783)
    d8 = GD.GaussianDiffusion(unet_m0, image_size=16, timesteps=8, beta_schedule="cosine", auto_normalize=False)
    torch.manual_seed(31)
    s = d8.sample(batch_size=2)
    torch.manual_seed(31)
    xT = torch.randn(2, 2, 16, 16)
    nzs = [torch.randn(2, 2, 16, 16) for _ in range(7)]
    g["loop8_out"], g["loop8_xT"], g["loop8_noises"] = n(s), n(xT), n(torch.stack(nzs))
    dd = GD.GaussianDiffusion(unet_m0, image_size=16, timesteps=8, sampling_timesteps=4, beta_schedule="cosine",
                              ddim_sampling_eta=1.0, auto_normalize=False)
    torch.manual_seed(32)
    s = dd.sample(batch_size=2)
    torch.manual_seed(32)
    xT = torch.randn(2, 2, 16, 16)
    nzs = [torch.randn(2, 2, 16, 16) for _ in range(3)]
    g["ddim_out"], g["ddim_xT"], g["ddim_noises"] = n(s), n(xT), n(torch.stack(nzs))
    g["meta"] = np.array(json.dumps(specs))
    np.savez_compressed(os.path.join(OUT, "diffusion.npz"), **g)
    print("diffusion.npz", len(g), "arrays")


@torch.no_grad()
def gen_unet_celeba():
    """The U-Net of BASELINE configs[4]'s first latent part (6x64x64, dim 64, dim_mults [1,2,2], groups 16; README.md:319-321):
    n = 4096 positions in the linear attention of the first stage (chunked core), 64x64 GroupNorm groups, full attention at 16x16."""
    GD = R["gd"]
    g = {}
    u = GD.Unet(dim=64, dim_mults=(1, 2, 2), channels=6, resnet_block_groups=16)
    spec = {"spec": load_synth(u, 27), "ch": 6, "size": 64, "mults": [1, 2, 2], "groups": 16}
    x = t(synth_input("unetC64", (1, 6, 64, 64), 27))
    tt = torch.tensor([421], dtype=torch.long)
    g["x"], g["t"], g["y"] = n(x), n(tt), n(u(x, tt))
    d = GD.GaussianDiffusion(u, image_size=64, timesteps=1000, beta_schedule="cosine", loss_type="l2", auto_normalize=False)
    eps = t(synth_input("unetC64eps", (1, 6, 64, 64), 27))
    g["eps"], g["p_losses"] = n(eps), n(d.p_losses(x * 0.8, tt, noise=eps))
    torch.manual_seed(421)
    img, xs = d.p_sample((x * 0.8).clone(), 421)
    torch.manual_seed(421)
    g["psample_noise"] = n(torch.randn_like(x))
    g["psample_img"], g["psample_x0"] = n(img), n(xs)
    g["meta"] = np.array(json.dumps(spec))
    np.savez_compressed(os.path.join(OUT, "unet_celeba.npz"), **g)
    print("unet_celeba.npz", len(g), "arrays, |y|max", float(np.abs(g["y"]).max()))


# ------------------------------------------------------------------ objectives pred_x0 / pred_v, self-conditioning, auto_normalize
@torch.no_grad()
def gen_diffusion_variants():
    """GaussianDiffusion options off the drivers' default path (gaussian_diffusion.py:647-663, :674-694, :726-728, :744-747, :832-855,
    :148-157): same U-Net shape as `m0` of diffusion.npz."""
    GD = R["gd"]
    g, meta = {}, {}
    x0 = t(synth_input("x0", (2, 2, 16, 16), 22)) * 0.8
    eps = t(synth_input("eps", (2, 2, 16, 16), 23))
    tt = torch.tensor([10, 700], dtype=torch.long)
    g["x0"], g["eps"], g["t"] = n(x0), n(eps), n(tt)

    def loops(tag, unet, **kw):
        d8 = GD.GaussianDiffusion(unet, image_size=16, timesteps=8, beta_schedule="cosine", auto_normalize=False, **kw)
        torch.manual_seed(31)
        s8 = d8.sample(batch_size=2)
        torch.manual_seed(31)
        xT = torch.randn(2, 2, 16, 16)
        nzs = [torch.randn(2, 2, 16, 16) for _ in range(7)]
        g[f"{tag}_loop8_out"], g[f"{tag}_loop8_xT"], g[f"{tag}_loop8_noises"] = n(s8), n(xT), n(torch.stack(nzs))
        dd = GD.GaussianDiffusion(unet, image_size=16, timesteps=8, sampling_timesteps=4, beta_schedule="cosine",
                                  ddim_sampling_eta=1.0, auto_normalize=False, **kw)
        torch.manual_seed(32)
        sd = dd.sample(batch_size=2)
        torch.manual_seed(32)
        xT = torch.randn(2, 2, 16, 16)
        nzs = [torch.randn(2, 2, 16, 16) for _ in range(3)]
        g[f"{tag}_ddim_out"], g[f"{tag}_ddim_xT"], g[f"{tag}_ddim_noises"] = n(sd), n(xT), n(torch.stack(nzs))

    u = GD.Unet(dim=64, dim_mults=(1, 2), channels=2, resnet_block_groups=8)
    meta["unet"] = load_synth(u, 21)
    for obj in ("pred_x0", "pred_v"):
        d = GD.GaussianDiffusion(u, image_size=16, timesteps=1000, beta_schedule="cosine", loss_type="l2", objective=obj,
                                 auto_normalize=False)
        g[f"{obj}_p_losses"] = n(d.p_losses(x0, tt, noise=eps))
        for clip in (False, True):
            mp = d.model_predictions(x0, tt, clip_x_start=clip)
            g[f"{obj}_mp_noise_clip{int(clip)}"], g[f"{obj}_mp_x0_clip{int(clip)}"] = n(mp.pred_noise), n(mp.pred_x_start)
        for step in (0, 500):
            torch.manual_seed(100 + step)
            img, xs = d.p_sample(x0.clone(), step)
            torch.manual_seed(100 + step)
            nz = torch.randn_like(x0)
            g[f"{obj}_psample{step}_img"], g[f"{obj}_psample{step}_x0"], g[f"{obj}_psample{step}_noise"] = n(img), n(xs), n(nz)
        g[f"{obj}_predict_v"] = n(d.predict_v(x0, tt, eps))
        g[f"{obj}_noise_from_start"] = n(d.predict_noise_from_start(x0, tt, eps))
        loops(obj, u, objective=obj)
    # self-conditioning (pred_noise): U-Net with 2*channels inputs
    usc = GD.Unet(dim=64, dim_mults=(1, 2), channels=2, resnet_block_groups=8, self_condition=True)
    meta["unet_sc"] = load_synth(usc, 24)
    sc = t(synth_input("selfcond", (2, 2, 16, 16), 25)) * 0.5
    g["sc"] = n(sc)
    g["sc_unet_given"] = n(usc(x0, tt, sc))
    g["sc_unet_none"] = n(usc(x0, tt))
    d = GD.GaussianDiffusion(usc, image_size=16, timesteps=1000, beta_schedule="cosine", loss_type="l2", auto_normalize=False)
    for name, draw in (("taken", 0.1), ("skipped", 0.9)):
        GD.random = lambda draw=draw: draw                               # `from random import random` at :5 of the reference module
        g[f"sc_p_losses_{name}"] = n(d.p_losses(x0, tt, noise=eps))
    torch.manual_seed(105)
    img, xs = d.p_sample(x0.clone(), 500, sc)
    torch.manual_seed(105)
    nz = torch.randn_like(x0)
    g["sc_psample500_img"], g["sc_psample500_x0"], g["sc_psample500_noise"] = n(img), n(xs), n(nz)
    loops("sc", usc)
    # auto_normalize=True: forward() maps [0,1] images to [-1,1], sample() maps back (gaussian_diffusion.py:148-157, :637-638)
    dan = GD.GaussianDiffusion(u, image_size=16, timesteps=8, beta_schedule="cosine", loss_type="l2", auto_normalize=True)
    x01 = (x0 * 0.5 + 0.5).clamp(0, 1)
    torch.manual_seed(9)
    loss = dan(x01)
    torch.manual_seed(9)
    t_drawn = torch.randint(0, 8, (2,)).long()
    nz = torch.randn_like(x0)
    g["an_x01"], g["an_fwd_loss"], g["an_fwd_t"], g["an_fwd_noise"] = n(x01), n(loss), n(t_drawn), n(nz)
    torch.manual_seed(31)
    g["an_loop8_out"] = n(dan.sample(batch_size=2))
    g["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "diffusion_variants.npz"), **g)
    print("diffusion_variants.npz", len(g), "arrays")


# ------------------------------------------------------------------ variational bound + CatFormater goldens
def gen_unet_embeddings():
    """U-Nets with random / learned sinusoidal time embeddings (gaussian_diffusion.py:240-255, :398-405): forward outputs, and for the
    learned variant the gradient of its frequencies and of the first time-MLP linear under loss = sum(out * eps)."""
    GD = R["gd"]
    g, meta = {}, {}
    x0 = t(synth_input("x0", (2, 2, 16, 16), 22)) * 0.8
    eps = t(synth_input("eps", (2, 2, 16, 16), 23))
    tt = torch.tensor([10, 700], dtype=torch.long)
    g["x0"], g["eps"], g["t"] = n(x0), n(eps), n(tt)
    for tag, kw in (("learned", dict(learned_sinusoidal_cond=True)), ("random", dict(random_fourier_features=True))):
        u = GD.Unet(dim=64, dim_mults=(1, 2), channels=2, resnet_block_groups=8, learned_sinusoidal_dim=16, **kw)
        meta[tag] = load_synth(u, 26)
        out = u(x0, tt)
        g[f"{tag}_out"] = n(out)
        g[f"{tag}_emb"] = n(u.time_mlp[0](tt))
        g[f"{tag}_requires_grad"] = np.array(int(u.time_mlp[0].weights.requires_grad))
        if u.time_mlp[0].weights.requires_grad:
            (out * eps).sum().backward()
            g[f"{tag}_grad_weights"] = n(u.time_mlp[0].weights.grad)
            g[f"{tag}_grad_lin1"] = n(u.time_mlp[1].weight.grad)
            g[f"{tag}_grad_init_conv"] = n(u.init_conv.weight.grad)
    g["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "unet_embeddings.npz"), **g)
    print("unet_embeddings.npz", len(g), "arrays")


@torch.no_grad()
def gen_vlb():
    GD, FM = R["gd"], R["formaters"]
    g = {}
    u = GD.Unet(dim=64, dim_mults=(1, 2), channels=2, resnet_block_groups=8)
    spec = {"spec": load_synth(u, 21), "ch": 2, "size": 16, "mults": [1, 2], "groups": 8}
    x0 = t(synth_input("x0", (2, 2, 16, 16), 22)) * 0.8
    eps = t(synth_input("eps", (2, 2, 16, 16), 23))
    g["x0"], g["eps"] = n(x0), n(eps)
    # single terms, T=1000 cosine: decoder NLL (t=0) and posterior KL (t>0) in one batch  (gaussian_diffusion.py:932-958)
    d = GD.GaussianDiffusion(u, image_size=16, timesteps=1000, beta_schedule="cosine", auto_normalize=False)
    for tag, tv in (("a", [0, 700]), ("b", [1, 999]), ("c", [0, 0])):
        tt = torch.tensor(tv, dtype=torch.long)
        xt = d.q_sample(x0, tt, eps)
        g[f"term_{tag}_t"], g[f"term_{tag}_xt"] = n(tt), n(xt)
        g[f"term_{tag}_out"] = n(d._vb_terms_bpd(x_start=x0, x_t=xt, t=tt))
    g["prior_T1000"] = n(d._prior_bpd(x0))
    # full loop, T=8: noise drawn by randn_like once per step, t = T-1 .. 0  (gaussian_diffusion.py:869-903)
    d8 = GD.GaussianDiffusion(u, image_size=16, timesteps=8, beta_schedule="cosine", auto_normalize=False)
    torch.manual_seed(41)
    total = d8.calc_neg_log_likelihood_loop(x0)
    torch.manual_seed(41)
    nzs = [torch.randn_like(x0) for _ in range(8)]
    g["loop8_total"], g["loop8_noises"], g["prior_T8"] = n(total), n(torch.stack(nzs)), n(d8._prior_bpd(x0))
    # CatFormater index maps (latent_formaters.py:135-244): integer-valued latents so equality is exact
    for tag, L, C, size, B in (("L3", 3, 3, 32, 2), ("L5", 5, 3, 64, 1), ("L4", 4, 1, 32, 2)):
        f = FM.CatFormater(L, C, size)
        lat = [torch.arange(B * int(np.prod(dm)), dtype=torch.float32).reshape(B, *[int(v) for v in dm]) + 1000.0 * i
               for i, dm in enumerate(f.latent_dims)]
        out = f.process_latents([l.clone() for l in lat])
        assert len(out) == 1
        back = f.postprocess([out[0].clone()])
        assert all(torch.equal(a, b) for a, b in zip(back, lat)), tag
        g[f"cat_{tag}_out"] = n(out[0])
        g[f"cat_{tag}_shape"] = np.array([int(v) for v in f.get_input_shapes()[0]])
        spec[f"cat_{tag}"] = {"L": L, "C": C, "size": size, "B": B}
    g["meta"] = np.array(json.dumps(spec))
    np.savez_compressed(os.path.join(OUT, "vlb.npz"), **g)
    print("vlb.npz", len(g), "arrays", {k: g[k].tolist() for k in ("term_a_out", "prior_T1000", "loop8_total")})


# ------------------------------------------------------------------ training-step goldens (trainer.py:152-167)
def gen_train(C=3, L=3, K=2, B=2, seed=2):
    """loss.backward() of the reference on the N-config golden weights: gradients of a spread of named parameters before
    and after clip_grad_value_(flow.parameters(), 1) + clip_grad_norm_(flow.parameters(), 1), the total norm, and the
    parameters after one Adam step (lr 1e-3) - the body of the reference's training loop."""
    U = R["nf_utils"]
    flow = R["glow"].Glow(in_channel=C, L=L, K=K)
    prior = R["prior"].GaussianPrior(2 ** (L + 1) * C)
    spec, pspec = load_synth(flow, seed), load_synth(prior, seed)
    x_np, _, _ = images_5bit("glowN", (B, C, 32, 32), seed)          # same inputs/weights as glow_N.npz
    x = t(x_np)
    opt = U.init_optimizer("adam", [{"params": flow.parameters()}, {"params": prior.parameters()}], 1e-3)
    ldj, logp = f64z(B), f64z(B)
    zs, ldj, logp = flow.transform(x, ldj, logp)
    logp = logp + prior.compute_log_prob(zs[-1])
    loss = U.calculate_loss(ldj + logp, 32.0, 32 * 32 * 3.0)
    opt.zero_grad()
    loss.backward()
    names = dict(list(flow.named_parameters()))
    pnames = dict(list(prior.named_parameters()))
    # a spread over levels, layer types and positions in the flow
    keys = [k for k in names if any(k.startswith(p) for p in ("blocks.0.flows.0.", "blocks.1.flows.1.", "final_flows.0.", "final_flows.1."))
            or ".split." in k]
    def summary(v):
        """[first 64 values | sum | sum of squares | max abs] in fp64: a fixture of a few hundred bytes per tensor."""
        f = v.detach().double().reshape(-1)
        return np.concatenate([f[:64].numpy(), [float(f.sum()), float((f * f).sum()), float(f.abs().max())]])

    g = {"bpd": n(loss), "keys": np.array(json.dumps(keys)), "pkeys": np.array(json.dumps(list(pnames)))}
    for i, k in enumerate(keys):
        g[f"g{i}"] = summary(names[k].grad)
    for i, k in enumerate(pnames):
        g[f"pg{i}"] = summary(pnames[k].grad)
    g["sumsq_all"] = np.array(sum(float((p.grad.double() ** 2).sum()) for p in flow.parameters()))
    torch.nn.utils.clip_grad_value_(flow.parameters(), 1)
    total = torch.nn.utils.clip_grad_norm_(flow.parameters(), 1)
    g["total_norm_after_value_clip"] = n(total)
    for i, k in enumerate(keys):
        g[f"c{i}"] = summary(names[k].grad)
    opt.step()
    for i, k in enumerate(keys):
        g[f"p{i}"] = summary(names[k])
    for i, k in enumerate(pnames):
        g[f"pp{i}"] = summary(pnames[k])
    g["meta"] = np.array(json.dumps({"C": C, "L": L, "K": K, "B": B, "seed": seed, "flow_spec": spec, "prior_spec": pspec}))
    np.savez_compressed(os.path.join(OUT, "train_N.npz"), **g)
    print(f"train_N.npz bpd={float(loss):.6f} total_norm={float(total):.4f} n_keys={len(keys)}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["ops", "glow", "celeba", "init", "diffusion", "variants", "embeddings", "vlb", "train"]
    if "ops" in which:
        gen_ops()
    if "glow" in which:
        gen_glow("M", 1, 3, 4, 4, 1)      # BASELINE configs[0] shape (MNIST 1x32x32, L3 K4)
        gen_glow("N", 3, 3, 2, 2, 2)      # north-star shape with reduced K
        gen_glow("D", 3, 5, 1, 2, 3)      # CelebA depth (BASELINE configs[4]: L = 5 -> C up to 192) on 32x32 inputs
    if "celeba" in which:
        gen_glow("C", 3, 5, 2, 1, 6, size=128, profile="DEEP")   # BASELINE configs[4] image size and depth (3x128x128, L5), K reduced to 2
        gen_unet_celeba()                        # ... and the U-Net of its first latent part (6x64x64, mults [1,2,2], groups 16)
    if "init" in which:
        gen_glow_init()
    if "diffusion" in which:
        gen_diffusion()
    if "variants" in which:
        gen_diffusion_variants()
    if "embeddings" in which:
        gen_unet_embeddings()
    if "vlb" in which:
        gen_vlb()
    if "train" in which:
        gen_train()
    leftovers = [r for r, dnames, _ in os.walk("/root/reference") if "__pycache__" in dnames]
    assert not leftovers, leftovers
