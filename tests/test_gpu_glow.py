"""GPU parity, composite: Glow.transform / invert / sample + GaussianPrior + bits/dim against the golden vectors
captured from the reference (configs M and N-reduced), the data-dependent init path, and size-independent
properties at BASELINE.json's full size (L3 K16, B=512): round trip, batch-slicing invariance, bf16-vs-fp32 bpd.
"""
import pytest
import torch

import nfdp_hip as K
from oracle import nf_oracle as O
from tests.util import load_golden, meta_of, synth_sd, T, assert_close, relerr, relerr_elem, record
from tests.golden.synth import synth_tensor, images_5bit, DEEP

pytestmark = pytest.mark.gpu
TOL = 1e-4          # BASELINE.json: fp32 bits/dim and sampled pixels within 1e-4 relative
DEV = "cuda"


def z64(b):
    return torch.zeros(b, dtype=torch.float64, device=DEV)


def build(m, compute="f32"):
    from normalizing_flow import Glow, GaussianPrior
    flow = Glow(in_channel=m["C"], L=m["L"], K=m["K"], compute=compute)
    prior = GaussianPrior(2 ** (m["L"] + 1) * m["C"])
    flow.load_state_dict(synth_sd(m["flow_spec"], m["seed"], profile=m.get("profile")), strict=True)
    prior.load_state_dict(synth_sd(m["prior_spec"], m["seed"], profile=m.get("profile")), strict=True)
    return flow, prior


@pytest.mark.parametrize("tag", ["M", "N", "D", "C"])
@torch.no_grad()
def test_glow_forward_nll_golden(tag):
    """Reference outputs (tools/make_goldens.py).  C = BASELINE configs[4]'s image size and depth: 3x128x128, L = 5 (K = 2), whose
    64x64 / 32x32 levels run `coupling_mlp_kernel` and whose 16x16 / 8x8 / 4x4 levels carry C = 48 / 96 / 192."""
    from normalizing_flow import calculate_loss
    g = load_golden(f"glow_{tag}.npz")
    m = meta_of(g)
    size = m.get("size", 32)
    flow, prior = build(m)
    x = T(g["x"]).to(DEV)
    assert x.shape[-1] == size
    ldj, logp = z64(m["B"]), z64(m["B"])
    zs, ldj_r, logp_r = flow.transform(x, ldj, logp)                     # trainer.py:155
    assert ldj_r is ldj and logp_r is logp                                # in-place accumulators, as the reference
    for i, z in enumerate(zs):
        assert tuple(z.shape) == g[f"z{i}"].shape
        assert_close(z, g[f"z{i}"], TOL, f"z{i}")
    assert_close(logp, g["logp_flow"], TOL, "logp_flow")
    logp += prior.compute_log_prob(zs[-1])                                # trainer.py:156
    assert_close(ldj, g["ldj"], TOL, "ldj")
    assert_close(logp, g["logp"], TOL, "logp")
    bpd = calculate_loss(ldj + logp, 32.0, size * size * 3.0)             # trainer.py:157-161
    assert_close(bpd, g["bpd"], TOL, "bpd")
    # NFBackbone path: logp=None
    zs2, _, none = flow.transform(x, z64(m["B"]), None)
    assert none is None and all(torch.equal(a, b) for a, b in zip(zs, zs2))


BF16_Z_TOL = 5e-2       # bf16 MFMA operands (8-bit mantissa), fp32 accumulation and flow state: latents / pixels at the 1e-2 level
BF16_BPD_TOL = 1e-4     # ... while bits/dim (a mean over ~3k log terms per image) stays inside BASELINE.json's gate


@pytest.mark.parametrize("tag", ["M", "N", "D", "C"])
@torch.no_grad()
def test_glow_bf16_mode_against_reference_goldens(tag):
    """The bf16 speed mode (fused flow-step kernels; `coupling_mlp_kernel` at the 64x64 / 32x32 levels of C) against the REFERENCE's
    outputs, forward + NLL and both inverses, at the separately stated bf16 tolerances."""
    from normalizing_flow import calculate_loss
    g = load_golden(f"glow_{tag}.npz")
    m = meta_of(g)
    size = m.get("size", 32)
    flow, prior = build(m, "bf16")
    x = T(g["x"]).to(DEV)
    ldj, logp = z64(m["B"]), z64(m["B"])
    zs, _, _ = flow.transform(x, ldj, logp)
    for i, z in enumerate(zs):
        assert_close(z, g[f"z{i}"], BF16_Z_TOL, f"bf16 z{i}")
    logp += prior.compute_log_prob(zs[-1])
    assert_close(ldj, g["ldj"], 1e-3, "bf16 ldj")
    assert_close(logp, g["logp"], 1e-3, "bf16 logp")
    assert_close(calculate_loss(ldj + logp, 32.0, size * size * 3.0), g["bpd"], BF16_BPD_TOL, "bf16 bpd")
    gz = [T(g[f"z{i}"]).to(DEV) for i in range(m["L"])]
    assert_close(flow.invert(gz), g["x_rec"], BF16_Z_TOL, "bf16 x_rec")
    eps = [T(g[f"eps{j}"]).to(DEV) for j in range(m["L"] - 1)]
    assert_close(flow.invert([gz[-1]], temperature=0.7, eps=eps), g["x_sampled"], BF16_Z_TOL, "bf16 x_sampled")


@pytest.mark.parametrize("tag", ["M", "N", "D", "C"])
@torch.no_grad()
def test_glow_inverse_and_sampling_golden(tag):
    from normalizing_flow import postprocess_batch
    g = load_golden(f"glow_{tag}.npz")
    m = meta_of(g)
    flow, _ = build(m)
    zs = [T(g[f"z{i}"]).to(DEV) for i in range(m["L"])]
    keep = [z.clone() for z in zs]
    assert_close(flow.invert(zs), g["x_rec"], TOL, "x_rec")
    assert all(torch.equal(a, b) for a, b in zip(zs, keep))              # caller's latents untouched
    eps = [T(g[f"eps{j}"]).to(DEV) for j in range(m["L"] - 1)]
    assert_close(flow.invert([zs[-1]], temperature=0.7, eps=eps), g["x_sampled"], TOL, "x_sampled")
    post = flow.sample([zs[-1]], postprocess_func=lambda v: postprocess_batch(v, 32.0), temperature=0.0)
    ref = T(g["x_sampled_post"])
    assert post.dtype == torch.uint8 and post.shape == ref.shape
    assert (post != ref).float().mean() < 1e-3                            # floor() bin edges only
    assert flow.sample([zs[-1]], temperature=0.7).shape == g["x_rec"].shape   # device RNG path


@torch.no_grad()
def test_glow_data_dependent_init_golden():
    from normalizing_flow import Glow, GaussianPrior, calculate_loss
    g = load_golden("glow_init.npz")
    m = meta_of(g)
    flow = Glow(in_channel=m["C"], L=m["L"], K=m["K"], compute="f32")
    sd = flow.state_dict()
    new = {k: (T(synth_tensor(k, v.shape, m["seed"])) if k.endswith(("_Conv2dActNorm__conv.weight", "invconv2d.weight"))
               else v) for k, v in sd.items()}
    flow.load_state_dict(new, strict=True)
    assert all(int(v) == 0 for k, v in flow.state_dict().items() if k.endswith("is_initialized"))
    prior = GaussianPrior(2 ** (m["L"] + 1) * m["C"])
    x = T(g["x"]).to(DEV)
    ldj, logp = z64(m["B"]), z64(m["B"])
    zs, _, _ = flow.transform(x, ldj, logp)
    logp += prior.compute_log_prob(zs[-1])
    for i, z in enumerate(zs):
        assert_close(z, g[f"z{i}"], TOL, f"z{i}")
    assert_close(ldj, g["ldj"], TOL, "ldj")
    assert_close(calculate_loss(ldj + logp, 32.0, 3072.0), g["bpd"], TOL, "bpd")
    after = flow.state_dict()
    assert all(int(v) == 1 for k, v in after.items() if k.endswith("is_initialized"))
    for k in [k for k in g if k.startswith("p:")]:
        assert_close(after[k[2:]], g[k], TOL, k)
    ldj2, logp2 = z64(m["B"]), z64(m["B"])
    zs2, _, _ = flow.transform(T(g["x2"]).to(DEV), ldj2, logp2)
    assert_close(zs2[-1], g["z2_last"], TOL, "z2_last")
    assert_close(ldj2, g["ldj2"], TOL, "ldj2")
    assert_close(logp2, g["logp2_flow"], TOL, "logp2")


def _north_star(compute, K_flows=16):
    from normalizing_flow import Glow, GaussianPrior
    flow = Glow(in_channel=3, L=3, K=K_flows, compute=compute)
    prior = GaussianPrior(48)
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 7, DEEP)) for k, v in flow.state_dict().items()},
                         strict=True)
    prior.load_state_dict({k: T(synth_tensor(k, v.shape, 7, DEEP)) for k, v in prior.state_dict().items()}, strict=True)
    return flow, prior


@torch.no_grad()
def test_north_star_full_size_properties():
    """L3 K16, B=512 (BASELINE configs[1]): encode->decode round trip, batch-slice invariance, determinism,
    and fp32-vs-oracle on a 4-image slice of the same batch (the oracle needs ~1 s/img on CPU)."""
    from normalizing_flow import calculate_loss
    flow, prior = _north_star("f32")
    x_np, _, _ = images_5bit("ns", (512, 3, 32, 32), 7)
    x = T(x_np).to(DEV)
    ldj, logp = z64(512), z64(512)
    zs, _, _ = flow.transform(x, ldj, logp)
    logp += prior.compute_log_prob(zs[-1])
    assert [tuple(z.shape[1:]) for z in zs] == [(6, 16, 16), (12, 8, 8), (48, 4, 4)]
    assert all(torch.isfinite(z).all() for z in zs) and torch.isfinite(ldj).all() and torch.isfinite(logp).all()
    # round trip (the reference's own invertibility check, EPS scaled to the tensor)
    assert relerr(flow.invert(zs), x) < 1e-3
    # determinism: same input -> bitwise same output
    ldj_b, logp_b = z64(512), z64(512)
    zs_b, _, _ = flow.transform(x, ldj_b, logp_b)
    assert all(torch.equal(a, b) for a, b in zip(zs, zs_b)) and torch.equal(ldj, ldj_b)
    # samples are independent: a slice of the batch gives the same rows
    l8, p8 = z64(8), z64(8)
    zs8, _, _ = flow.transform(x[100:108], l8, p8)
    # (to fp32 summation order, not bitwise: an 8-image batch takes the split-K conv kernel, the 512-image batch the tiled one)
    assert relerr(zs8[-1], zs[-1][100:108]) < 5e-6 and relerr(l8, ldj[100:108]) < 1e-7
    # oracle on a 4-image slice
    sd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
    psd = {k: v.detach().cpu() for k, v in prior.state_dict().items()}
    zo, ldj_o, logp_o, bpd_o = O.fwd_nll(T(x_np[:4]), sd, psd, 3, 16)
    for a, b in zip(zs, zo):
        assert_close(a[:4], b, TOL, "z vs oracle")
    assert_close(ldj[:4], ldj_o, TOL, "ldj vs oracle")
    assert_close(logp[:4], logp_o, TOL, "logp vs oracle")
    bpd4 = calculate_loss((ldj + logp)[:4].contiguous(), 32.0, 3072.0)
    assert_close(bpd4, bpd_o, TOL, "bpd vs oracle")
    # inverse + Split sampling (sampled pixels) on the same slice with the same injected noise, against the oracle evaluated in fp64
    # (the fp32 oracle's own distance from it is recorded beside the HIP path's): within BASELINE.json's 1e-4, element-wise.
    sd64 = {k: v.double() for k, v in sd.items()}
    z_in = [z[:4].cpu().contiguous() for z in zs]
    eps = [torch.randn(4, 12, 8, 8), torch.randn(4, 6, 16, 16)]
    for what, lat, kw in (("x_rec", z_in, {}), ("sampled pixels", [z_in[-1]], {"temperature": 0.7, "eps": eps})):
        truth = O.glow_inv([z.double() for z in lat], sd64, 3, 16, kw.get("temperature", 1.0),
                           [e.double() for e in eps] if kw else None)
        ref32 = O.glow_inv(lat, sd, 3, 16, kw.get("temperature", 1.0), eps if kw else None)
        hip = flow.invert([z.to(DEV) for z in lat], **({"temperature": 0.7, "eps": [e.to(DEV) for e in eps]} if kw else {}))
        e_hip, e_ref = relerr_elem(hip, truth), relerr_elem(ref32, truth)
        record(f"north_star_f32_{what.replace(' ', '_')}_vs_fp64_truth", e_hip)
        record(f"north_star_f32_oracle_{what.replace(' ', '_')}_vs_fp64_truth", e_ref)
        # measured on MI355X (profiles/r03_parity_metrics.json): 8.6e-6 / 3.2e-6 for the HIP path, 1.1e-5 / 3.4e-6 for the fp32 oracle
        assert e_hip <= TOL, f"{what}: hip {e_hip:.2e} vs fp64 truth (fp32 oracle itself: {e_ref:.2e})"


@torch.no_grad()
def test_north_star_split_product_mode_meets_the_pixel_gate():
    """compute="f32" with K.set_f32_coupling("bf16x3") (every level on nfdp_coupling_mlp_gather_bf16x3: 16x16x12 and 8x8x24 on the
    128-pixel kernel, 4x4x48 on the row-split one): L3 K16, B = 512 - bits/dim inside the 1e-4 gate against the oracle, inverse and sampled pixels within 1e-4 of the
    oracle evaluated in fp64 (element-wise), determinism, and the split kernel really ran."""
    from normalizing_flow import calculate_loss
    flow, prior = _north_star("f32")
    x_np, _, _ = images_5bit("ns", (512, 3, 32, 32), 7)
    x = T(x_np).to(DEV)
    K.set_f32_coupling("bf16x3")
    try:
        ldj, logp = z64(512), z64(512)
        K.start_trace()
        zs, _, _ = flow.transform(x, ldj, logp)
        labels = [e["label"] for e in K.stop_trace()]
        assert sum(lb.startswith("coupling_mlp_x3") for lb in labels) == 48, labels      # all 16 flows of all three levels
        logp += prior.compute_log_prob(zs[-1])
        ldj_b, logp_b = z64(512), z64(512)
        zs_b, _, _ = flow.transform(x, ldj_b, logp_b)
        assert all(torch.equal(a, b) for a, b in zip(zs, zs_b)) and torch.equal(ldj, ldj_b)
        sd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
        psd = {k: v.detach().cpu() for k, v in prior.state_dict().items()}
        zo, ldj_o, logp_o, bpd_o = O.fwd_nll(T(x_np[:4]), sd, psd, 3, 16)
        bpd4 = calculate_loss((ldj + logp)[:4].contiguous(), 32.0, 3072.0)
        record("north_star_f32x3_bpd_vs_oracle", relerr(bpd4, bpd_o))
        assert_close(bpd4, bpd_o, TOL, "bpd vs oracle (split products)")
        for i, (a, b) in enumerate(zip(zs, zo)):
            record(f"north_star_f32x3_z{i}_vs_oracle", relerr_elem(a[:4], b))
            assert_close(a[:4], b, 1e-3, f"z{i} vs oracle (split products)")
        sd64 = {k: v.double() for k, v in sd.items()}
        z_in = [z[:4].cpu().contiguous() for z in zs]
        g = torch.Generator().manual_seed(5)
        eps = [torch.randn(4, 12, 8, 8, generator=g), torch.randn(4, 6, 16, 16, generator=g)]
        for what, lat, kw in (("x_rec", z_in, {}), ("sampled pixels", [z_in[-1]], {"temperature": 0.7, "eps": eps})):
            truth = O.glow_inv([z.double() for z in lat], sd64, 3, 16, kw.get("temperature", 1.0), [e.double() for e in eps] if kw else None)
            hip = flow.invert([z.to(DEV) for z in lat], **({"temperature": 0.7, "eps": [e.to(DEV) for e in eps]} if kw else {}))
            e_hip = relerr_elem(hip, truth)
            record(f"north_star_f32x3_{what.replace(' ', '_')}_vs_fp64_truth", e_hip)
            assert e_hip <= TOL, f"{what}: {e_hip:.2e} vs fp64 truth"
    finally:
        K.set_f32_coupling("mfma_f32")


BF16_DEEP_Z_TOL = 5e-2     # 48 flows of bf16-operand networks: latents / pixels (fp32 state, 8-bit-mantissa network inputs)


@torch.no_grad()
def test_north_star_bf16_benched_path_against_oracle():
    """The path `bench.py` times - bf16 fused level kernels, L3 K16, B = 512 (BASELINE configs[1]) - compared DIRECTLY with the
    oracle on image slices of the same batch (first and last four: both ends of the grid): per-sample ldj / logp at 1e-3, bits/dim
    inside BASELINE.json's 1e-4 gate, latents at the stated bf16 tolerance; and the bf16 inverse and sampling inverse at K = 16
    against the oracle evaluated in fp64."""
    from normalizing_flow import calculate_loss
    flow, prior = _north_star("bf16")
    x_np, _, _ = images_5bit("ns", (512, 3, 32, 32), 7)
    x = T(x_np).to(DEV)
    ldj, logp = z64(512), z64(512)
    zs, _, _ = flow.transform(x, ldj, logp)
    logp += prior.compute_log_prob(zs[-1])
    sd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
    psd = {k: v.detach().cpu() for k, v in prior.state_dict().items()}
    for lo in (0, 508):
        sl = slice(lo, lo + 4)
        zo, ldj_o, logp_o, bpd_o = O.fwd_nll(T(x_np[sl]), sd, psd, 3, 16)
        for i, (a, b) in enumerate(zip(zs, zo)):
            record(f"north_star_bf16_z{i}_vs_oracle_{lo}", relerr_elem(a[sl], b))
            assert_close(a[sl], b, BF16_DEEP_Z_TOL, f"bf16 z{i} vs oracle")
        assert_close(ldj[sl], ldj_o, 1e-3, "bf16 ldj vs oracle")
        assert_close(logp[sl], logp_o, 1e-3, "bf16 logp vs oracle")
        bpd4 = calculate_loss((ldj + logp)[sl].contiguous(), 32.0, 3072.0)
        record(f"north_star_bf16_bpd_vs_oracle_{lo}", relerr(bpd4, bpd_o))
        assert_close(bpd4, bpd_o, BF16_BPD_TOL, "bf16 bpd vs oracle")
    # inverse with all latents / with Split sampling, inputs = the ORACLE's latents of the first four images, truth = fp64 oracle
    sd64 = {k: v.double() for k, v in sd.items()}
    zo, _, _, _ = O.fwd_nll(T(x_np[:4]), sd, psd, 3, 16)
    eps = [torch.randn(4, 12, 8, 8, generator=torch.Generator().manual_seed(3)),
           torch.randn(4, 6, 16, 16, generator=torch.Generator().manual_seed(4))]
    # the batch the kernels see is the full 512 (rows 4.. are this run's own latents): the benched grid, not a 4-image launch
    for what, use_eps in (("x_rec", False), ("sampled pixels", True)):
        lat_o = [z.clone() for z in zo] if not use_eps else [zo[-1].clone()]
        truth = O.glow_inv([z.double() for z in lat_o], sd64, 3, 16, 0.7 if use_eps else 1.0,
                           [e.double() for e in eps] if use_eps else None)
        full = [z.clone() for z in zs] if not use_eps else [zs[-1].clone()]
        for f, o in zip(full, lat_o):
            f[:4] = o.to(DEV)
        kw = {}
        if use_eps:
            big = [torch.randn(512, *e.shape[1:], generator=torch.Generator().manual_seed(9)).to(DEV) for e in eps]
            for bg, e in zip(big, eps):
                bg[:4] = e.to(DEV)
            kw = {"temperature": 0.7, "eps": big}
        hip = flow.invert(full, **kw)[:4]
        e = relerr_elem(hip, truth)
        record(f"north_star_bf16_{what.replace(' ', '_')}_vs_fp64_truth", e)
        assert e <= BF16_DEEP_Z_TOL, f"bf16 {what}: {e:.2e} vs the fp64 oracle"


@torch.no_grad()
def test_bf16_mode_bits_per_dim_within_gate():
    """bf16 MFMA (fp32 accumulate) speed mode: bits/dim stays within the 1e-4 relative gate of the fp32 path;
    latents are compared at a separately stated looser tolerance (bf16 has an 8-bit mantissa)."""
    from normalizing_flow import calculate_loss
    x_np, _, _ = images_5bit("ns", (64, 3, 32, 32), 7)
    x = T(x_np).to(DEV)
    out = {}
    for mode in ("f32", "bf16"):
        flow, prior = _north_star(mode)
        ldj, logp = z64(64), z64(64)
        zs, _, _ = flow.transform(x, ldj, logp)
        logp += prior.compute_log_prob(zs[-1])
        out[mode] = (zs, calculate_loss(ldj + logp, 32.0, 3072.0))
    assert_close(out["bf16"][1], out["f32"][1], TOL, "bf16 bpd")
    assert relerr(out["bf16"][0][-1], out["f32"][0][-1]) < 5e-2


@torch.no_grad()
def test_nfbackbone_and_empty_batch():
    from normalizing_flow import NFBackbone
    nf = NFBackbone(None, in_channel=1, L=3, K=2, learn_prior_mean_logs=True, freeze_flow=True)
    assert nf.is_frozen() and all(not p.requires_grad for p in nf.parameters())
    x = torch.rand(4, 1, 32, 32, device=DEV) - 0.5
    lat, ldj = nf.transform(x, z64(4))
    assert len(lat) == 3 and ldj.shape == (4,)
    assert relerr(nf.invert(lat), x) < 1e-3
    assert nf.sample(lat).shape == x.shape


def test_grad_mode_selects_training_path_and_invert_still_raises():
    """With trainable parameters and grad mode on, transform runs the differentiable path (tests/test_gpu_backward.py);
    invert / sample have no backward and must say so instead of silently dropping the graph."""
    from normalizing_flow import Glow
    flow = Glow(in_channel=1, L=2, K=1, compute="f32")
    zs, ldj, logp = flow.transform(torch.rand(2, 1, 8, 8, device=DEV) - 0.5, z64(2), z64(2))
    assert ldj.requires_grad and logp.requires_grad and zs[-1].requires_grad
    with pytest.raises(NotImplementedError):
        flow.invert([z.detach() for z in zs])
    with torch.no_grad():
        flow.invert([z.detach() for z in zs])


@pytest.mark.parametrize("compute,tol", [("f32", TOL), ("bf16", 3e-2)])
@torch.no_grad()
def test_five_level_model_against_oracle(compute, tol):
    """The CelebA-depth model (BASELINE configs[4]: L=5, 3 channels -> C = 12 .. 192 per level) on 32x32 inputs, K=1: the
    channel counts beyond the 32x32 models' (C = 96, 192: templated per-pixel product, Gauss-Jordan log-dets from a global
    workspace, 512 -> 192 zero conv) against the oracle, forward + NLL and the inverse."""
    from normalizing_flow import Glow, GaussianPrior
    flow = Glow(in_channel=3, L=5, K=1, compute=compute)
    prior = GaussianPrior(192)
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 9, DEEP)) for k, v in flow.state_dict().items()}, strict=True)
    prior.load_state_dict({k: T(synth_tensor(k, v.shape, 9, DEEP)) for k, v in prior.state_dict().items()}, strict=True)
    flow.to(DEV); prior.to(DEV)
    sd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
    psd = {k: v.detach().cpu() for k, v in prior.state_dict().items()}
    x_np, _, _ = images_5bit("deep", (3, 3, 32, 32), 9)
    x = T(x_np).to(DEV)
    ldj, logp = z64(3), z64(3)
    zs, _, _ = flow.transform(x, ldj, logp)
    logp += prior.compute_log_prob(zs[-1])
    assert [tuple(z.shape[1:]) for z in zs] == [(6, 16, 16), (12, 8, 8), (24, 4, 4), (48, 2, 2), (192, 1, 1)]
    zo, ldj_o, logp_o, _ = O.fwd_nll(T(x_np), sd, psd, 5, 1)
    for i, (a, b) in enumerate(zip(zs, zo)):
        assert_close(a, b, tol, f"z{i}")
    assert_close(ldj, ldj_o, tol, "ldj")
    assert_close(logp, logp_o, tol, "logp")
    assert_close(flow.invert(zs), x, 1e-3 if compute == "f32" else 5e-2, "round trip")


@pytest.mark.parametrize("c,L,kf,size,b", [(1, 5, 1, 32, 3), (3, 4, 1, 64, 2), (1, 4, 2, 64, 2), (3, 2, 2, 16, 5), (1, 3, 2, 8, 4)])
@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_model_geometries_against_oracle(c, L, kf, size, b, compute):
    """Geometries beyond the two benchmarked models (1- and 3-channel data, 2 to 5 levels, 8x8 to 64x64 inputs, odd batches):
    forward + NLL against the oracle, inverse round trip, and one backward pass through the HIP training path (finite
    gradients for every parameter)."""
    from normalizing_flow import Glow, GaussianPrior, calculate_loss
    flow = Glow(in_channel=c, L=L, K=kf, compute=compute).to(DEV)
    prior = GaussianPrior(2 ** (L + 1) * c).to(DEV)
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 3, DEEP)) for k, v in flow.state_dict().items()})
    prior.load_state_dict({k: T(synth_tensor(k, v.shape, 3, DEEP)) for k, v in prior.state_dict().items()})
    x = torch.rand(b, c, size, size, generator=torch.Generator().manual_seed(c * 100 + L)) - 0.5
    tol = TOL if compute == "f32" else 2e-2
    with torch.no_grad():
        ldj, logp = z64(b), z64(b)
        zs, _, _ = flow.transform(x.to(DEV), ldj, logp)
        logp = logp + prior.compute_log_prob(zs[-1])
        sd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
        psd = {k: v.detach().cpu() for k, v in prior.state_dict().items()}
        zo, ldj_o, logp_o, _ = O.fwd_nll(x, sd, psd, L, kf)
        for i, (a, r) in enumerate(zip(zs, zo)):
            assert_close(a, r, tol, f"z{i}")
        assert_close(ldj, ldj_o, tol, "ldj")
        assert_close(logp, logp_o, tol, "logp")
        assert_close(flow.invert(zs), x, 1e-3 if compute == "f32" else 2e-2, "round trip")
    ldj, logp = z64(b), z64(b)
    zs, ldj, logp = flow.transform(x.to(DEV), ldj, logp)
    loss = calculate_loss(ldj + logp + prior.compute_log_prob(zs[-1]), 32.0, size * size * 3.0)
    loss.backward()
    for name, p in list(flow.named_parameters()) + list(prior.named_parameters()):
        assert p.grad is None or torch.isfinite(p.grad).all(), name


@torch.no_grad()
def test_graphed_forward_matches_eager():
    """nfdp_hip.graphs.Graphed: the forward + NLL step captured once and replayed on new batches equals the eager calls bitwise."""
    from nfdp_hip.graphs import Graphed
    flow, prior = _north_star("bf16", K_flows=2)
    flow.to(DEV); prior.to(DEV)

    def step(x):
        ldj, logp = z64(x.shape[0]), z64(x.shape[0])
        zs, ldj, logp = flow.transform(x, ldj, logp)
        return zs[-1], ldj + logp + prior.compute_log_prob(zs[-1])

    xs = [T(images_5bit(f"g{i}", (8, 3, 32, 32), 11 + i)[0]).to(DEV) for i in range(3)]
    graphed = Graphed(step, [xs[0]])
    for x in xs:
        z_ref, ll_ref = step(x)
        z, ll = graphed(x)
        assert torch.equal(z, z_ref) and torch.equal(ll, ll_ref)


@torch.no_grad()
def test_two_deep_transforms_on_two_streams_in_one_graph():
    """Two Glow.transform_nhwc + invert_nhwc passes of an L = 5 model issued on two streams inside ONE capture, replayed.  Round 4 died
    in hipStreamEndCapture on exactly this: ROCm's capture cannot end when it holds a NESTED fork (a stream forked from a forked
    stream - pure torch reproduces it, tools/probe_nested_capture.py, profiles/r05_b_nested_capture.txt), and each pass forked a side
    stream for its 1x1-conv log-dets / inverses.  Under capture the passes now fork only from the capture's origin stream
    (nfdp_hip.graphs.side_branch_allowed) and issue the side work inline elsewhere; results equal the eager passes bit for bit."""
    from normalizing_flow import Glow, glow as G
    from nfdp_hip import graphs
    assert G.LOGDETS_ON_SIDE_STREAM
    flow = Glow(in_channel=3, L=5, K=2, compute="bf16")
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 9, DEEP)) for k, v in flow.state_dict().items()}, strict=True)
    flow.to(DEV)
    xs = [K.to_nhwc(T(images_5bit(f"two{i}", (2, 3, 64, 64), 21 + i)[0]).to(DEV)).contiguous() for i in range(2)]
    ref = []
    for x in xs:                                               # eager, one after the other, on the current stream
        ldj = z64(2)
        zs = flow.transform_nhwc(x, ldj, None)
        ref.append(([z.clone() for z in zs], ldj.clone(), flow.invert_nhwc(zs).clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ldjs = [z64(2), z64(2)]
    out = [None, None]
    allowed = []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        allowed.append(graphs.side_branch_allowed())           # torch.cuda.graph's own capture stream: the origin
        for i, s in enumerate(streams):
            s.wait_stream(cap)
            with torch.cuda.stream(s):
                allowed.append(graphs.side_branch_allowed())   # a forked stream: no second-level fork
                ldjs[i].zero_()
                zs = flow.transform_nhwc(xs[i], ldjs[i], None)
                out[i] = (zs, flow.invert_nhwc(zs))
        for s in streams:
            cap.wait_stream(s)
    assert allowed == [True, False, False]
    assert not any(k[1] in {s.cuda_stream for s in streams} for k in G._SIDE_STREAMS)      # nothing forked from the forked streams
    for _ in range(2):
        g.replay()
        torch.cuda.synchronize()
        for i in range(2):
            zs_ref, ldj_ref, x_ref = ref[i]
            assert all(torch.equal(a, b) for a, b in zip(out[i][0], zs_ref))
            assert torch.equal(ldjs[i], ldj_ref) and torch.equal(out[i][1], x_ref)


@torch.no_grad()
def test_deep_model_capture_on_a_declared_origin_stream_keeps_its_side_branch():
    """torch.cuda.graph(g, stream=s): the origin stream is the caller's; declared with graphs.capture_origin(s) the pass forks its
    log-det side branch as in an ordinary capture, undeclared it runs them inline.  Both replay to the eager result."""
    from normalizing_flow import Glow, glow as G
    from nfdp_hip import graphs
    flow = Glow(in_channel=3, L=4, K=1, compute="bf16")
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 9, DEEP)) for k, v in flow.state_dict().items()}, strict=True)
    flow.to(DEV)
    x = K.to_nhwc(T(images_5bit("decl", (2, 3, 32, 32), 33)[0]).to(DEV)).contiguous()
    ldj_ref = z64(2)
    zs_ref = [z.clone() for z in flow.transform_nhwc(x, ldj_ref, None)]
    torch.cuda.synchronize()
    for declare in (True, False):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        ldj = z64(2)
        g = torch.cuda.CUDAGraph()
        with (graphs.capture_origin(s) if declare else __import__("contextlib").nullcontext()):
            with torch.cuda.graph(g, stream=s):
                assert graphs.side_branch_allowed() == declare
                ldj.zero_()
                zs = flow.transform_nhwc(x, ldj, None)
        assert ((x.device.index, s.cuda_stream) in G._SIDE_STREAMS) == declare
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(zs, zs_ref)) and torch.equal(ldj, ldj_ref)


@torch.no_grad()
def test_forward_and_inverse_aliases():
    """BASELINE.json's north_star names the surface Glow.forward() / inverse(); the reference only has transform / invert
    (glow.py:172, :203).  Both spellings reach the same code."""
    flow, _ = _north_star("f32", K_flows=1)
    flow.to(DEV)
    x = T(images_5bit("alias", (2, 3, 32, 32), 5)[0]).to(DEV)
    zs, ldj, logp = flow.transform(x, z64(2), z64(2))
    zs2, ldj2, logp2 = flow(x, z64(2), z64(2))
    zs3, _, _ = flow.forward(x, z64(2), z64(2))
    assert all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(zs, zs2, zs3))
    assert torch.equal(ldj, ldj2) and torch.equal(logp, logp2)
    assert torch.equal(flow.inverse(zs), flow.invert(zs))
    assert torch.equal(flow.inverse(zs[-1:], temperature=0.0), flow.invert(zs[-1:], temperature=0.0))


@torch.no_grad()
def test_reference_written_checkpoint_reproduces_the_reference_outputs():
    """The reference's own checkpoint file (tests/golden/ref_ckpt, tools/make_ref_checkpoint.py) loaded through load_model, run on the
    HIP path: latents, log-det, log-prob, bits/dim and the inverse equal what the reference computed with those weights."""
    import json
    import os
    import numpy as np
    from normalizing_flow import Glow, GaussianPrior, load_model, load_architecture, calculate_loss
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ckpt")
    g = np.load(os.path.join(d, "ref_ckpt.npz"))
    meta, arch = json.loads(str(g["meta"])), load_architecture(d)
    flow = Glow(in_channel=meta["C"], compute="f32", **arch)
    prior = GaussianPrior(2 ** (arch["L"] + 1) * meta["C"], arch["learn_prior_mean_logs"])
    load_model(flow, prior, d, meta["epoch"])
    flow.to(DEV); prior.to(DEV)
    b = meta["B"]
    x = T(g["x"]).to(DEV)
    ldj, logp = z64(b), z64(b)
    zs, ldj, logp = flow.transform(x, ldj, logp)
    logp = logp + prior.compute_log_prob(zs[-1])
    for i, z in enumerate(zs):
        assert_close(z, g[f"z{i}"], TOL, f"z{i}")
    assert_close(ldj, g["ldj"], TOL, "ldj")
    assert_close(logp, g["logp"], TOL, "logp")
    assert_close(calculate_loss(ldj + logp, 32.0, meta["size"] ** 2 * 3.0), g["bpd"], TOL, "bits/dim")
    assert_close(flow.invert(zs), g["x_rec"], 1e-3, "inverse")


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("c,L,size,b", [(3, 3, 32, 5), (1, 3, 32, 4), (3, 4, 64, 2), (1, 2, 16, 3)])
@torch.no_grad()
def test_fused_level_boundary_against_the_per_op_launches(compute, c, L, size, b):
    """glow.FUSED_LEVEL_BOUNDARY: Split (prior conv + log-density + latent copy) + the next Squeeze as one launch per boundary
    (csrc/boundary.hip: fp32 FMA chains for the narrow prior conv) and the levels' weight-only log-det terms in one launch, against
    the per-op launches: latents and log-det bit-identical (copies / the same fp64 adds), logp equal to the accuracy of the per-op
    path's prior conv (fp32 mode: 1e-6; bf16 mode: that conv runs on bf16 operands there), and both against the oracle."""
    from normalizing_flow import Glow, GaussianPrior, glow as G
    flow = Glow(in_channel=c, L=L, K=2, compute=compute).to(DEV)
    prior = GaussianPrior(2 ** (L + 1) * c).to(DEV)
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 13, DEEP)) for k, v in flow.state_dict().items()})
    prior.load_state_dict({k: T(synth_tensor(k, v.shape, 14, DEEP)) for k, v in prior.state_dict().items()})
    x = (torch.rand(b, c, size, size, generator=torch.Generator().manual_seed(7 * c + L)) - 0.5).to(DEV)
    out = {}
    assert G.FUSED_LEVEL_BOUNDARY
    min_batch = G.FUSED_BOUNDARY_MIN_BATCH
    G.FUSED_BOUNDARY_MIN_BATCH = 1                          # (the default keeps the fused launch for batches that fill the chip)
    try:
        for fused in (True, False):
            G.FUSED_LEVEL_BOUNDARY = fused
            K.start_trace()
            ldj, logp = z64(b), z64(b)
            zs, _, _ = flow.transform(x, ldj, logp)
            prior.accumulate_log_prob_nhwc(K.to_nhwc(zs[-1]), logp)
            out[fused] = (zs, ldj, logp, len(K.stop_trace()))
    finally:
        G.FUSED_LEVEL_BOUNDARY = True
        G.FUSED_BOUNDARY_MIN_BATCH = min_batch
    (zf, lf, pf, nf), (zu, lu, pu, nu) = out[True], out[False]
    assert all(torch.equal(a, b_) for a, b_ in zip(zf, zu)) and torch.equal(lf, lu)
    assert_close(pf, pu, 1e-6 if compute == "f32" else 2e-4, "logp, fused boundary vs per-op launches")
    sd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
    psd = {k: v.detach().cpu() for k, v in prior.state_dict().items()}
    _, ldj_o, logp_o, _ = O.fwd_nll(x.cpu(), sd, psd, L, 2)
    assert_close(pf, logp_o, TOL if compute == "f32" else 2e-3, "logp vs oracle")
    assert_close(lf, ldj_o, TOL if compute == "f32" else 2e-2, "ldj vs oracle")


@torch.no_grad()
def test_config4_full_size_properties():
    """BASELINE configs[4] at its full size - Glow L = 5, K = 16 on 128x128x3, batch 32 (the shape `legs.celeba` times: 16-flow launches
    of the wide levels, tiled coupling networks with the flow head in the tap pass, side-stream log-dets / inverses, nfdp_invconv_
    inverses, row-split tiles at 8192 pixels): bf16 bits/dim within 1e-4 (relative) of the fp32-mode path; the first two images
    against the CPU oracle (fp32 mode: 1e-4 on bits/dim, log-det, log-prob; bf16 mode: latents at bf16 accuracy); inverse round trip;
    batch-slicing invariance (to rounding: kernel forms follow the pixel count); bitwise run-to-run determinism; finite sampling inverse."""
    from normalizing_flow import Glow, GaussianPrior
    import numpy as np
    B, S = 32, 128
    sdf = None
    res = {}
    x_np, _, _ = images_5bit("celeba_full", (B, 3, S, S), 4)
    x = T(x_np).to(DEV)
    for compute in ("f32", "bf16"):
        flow = Glow(in_channel=3, L=5, K=16, compute=compute)
        prior = GaussianPrior(192)
        flow.load_state_dict({k: T(synth_tensor(k, v.shape, 1, DEEP)) for k, v in flow.state_dict().items()}, strict=True)
        prior.load_state_dict({k: T(synth_tensor(k, v.shape, 2, DEEP)) for k, v in prior.state_dict().items()}, strict=True)
        if sdf is None:
            sdf = ({k: v.detach().cpu().clone() for k, v in flow.state_dict().items()}, {k: v.detach().cpu().clone() for k, v in prior.state_dict().items()})
        flow.to(DEV); prior.to(DEV)

        def run(xb):
            ldj, logp = z64(xb.shape[0]), z64(xb.shape[0])
            zs, _, _ = flow.transform(xb, ldj, logp)
            logp = logp + prior.compute_log_prob(zs[-1])
            return zs, ldj, logp
        zs, ldj, logp = run(x)
        bpd = (-(ldj + logp) / (np.log(2.0) * S * S * 3) + 5.0)
        assert [tuple(z.shape[1:]) for z in zs] == [(6, 64, 64), (12, 32, 32), (24, 16, 16), (48, 8, 8), (192, 4, 4)]
        assert all(bool(torch.isfinite(z).all()) for z in zs) and bool(torch.isfinite(bpd).all())
        zs2, ldj2, logp2 = run(x)                                         # determinism
        assert all(torch.equal(a, b_) for a, b_ in zip(zs, zs2)) and torch.equal(ldj, ldj2) and torch.equal(logp, logp2)
        # a slice of the batch gives the slice of the result - to rounding, not bitwise: the kernel FORM follows the pixel count (3 images:
        # row-split tiles / one-wave fp32 tiles where 32 images take the 128-pixel / 16x16x32 forms; same operands, another fp32
        # summation order, and in bf16 mode the odd flipped rounding of a hidden activation)
        zs3, ldj3, logp3 = run(x[5:8])
        for i, (a, b_) in enumerate(zip(zs, zs3)):
            assert_close(b_, a[5:8], 2e-5 if compute == "f32" else 6e-2, f"{compute} z{i} of a batch slice")
        assert_close(ldj3, ldj[5:8], 1e-6 if compute == "f32" else 2e-3, f"{compute} log-det of a batch slice")
        assert_close(logp3, logp[5:8], 1e-6 if compute == "f32" else 2e-3, f"{compute} log-prob of a batch slice")
        rt = flow.invert(zs)
        assert_close(rt, x, 2e-3 if compute == "f32" else 0.15, f"{compute} round trip")      # (bf16 operands through 2 x 80 flows)
        record(f"config4_{compute}_round_trip", relerr_elem(rt, x))
        smp = flow.invert(zs[-1:], temperature=0.7)
        assert smp.shape == x.shape and bool(torch.isfinite(smp).all())
        res[compute] = (zs, ldj, logp, bpd)
        record(f"config4_{compute}_bpd_mean", float(bpd.mean()))
        del flow, prior
        torch.cuda.empty_cache()
    rel = float((res["bf16"][3].mean() - res["f32"][3].mean()).abs() / res["f32"][3].mean().abs())
    record("config4_bf16_bpd_vs_f32_mode", rel)
    assert rel < 1e-4, rel
    zo, ldj_o, logp_o, _ = O.fwd_nll(T(x_np[:2]), sdf[0], sdf[1], 5, 16)
    bpd_o = (-(ldj_o + logp_o) / (np.log(2.0) * S * S * 3) + 5.0)
    assert_close(res["f32"][3][:2], bpd_o, TOL, "fp32-mode bits/dim vs oracle")
    assert_close(res["f32"][1][:2], ldj_o, TOL, "fp32-mode log-det vs oracle")
    assert_close(res["f32"][2][:2], logp_o, TOL, "fp32-mode log-prob vs oracle")
    assert_close(res["bf16"][3][:2], bpd_o, TOL, "bf16 bits/dim vs oracle")
    for i, (a, b_) in enumerate(zip(res["f32"][0], zo)):
        assert_close(a[:2], b_, 2e-4, f"fp32-mode z{i} vs oracle")
    for i, (a, b_) in enumerate(zip(res["bf16"][0], zo)):
        assert_close(a[:2], b_, 6e-2, f"bf16 z{i} vs oracle")
