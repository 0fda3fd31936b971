"""Fused coupling-layer kernel (csrc/coupling_fused.hip) vs the unfused HIP path in the same bf16 mode (identical
rounding points -> agreement to summation order, occasionally one bf16 ulp on a hidden unit) and vs the fp32 CPU
oracle, for every supported geometry, ragged batch tails (partial last workgroup), forward and inverse."""
import pytest
import torch

import nfdp_hip as K
from oracle import nf_oracle as O
from tests.util import T, assert_close, relerr, record
from tests.golden.synth import synth_tensor, synth_input, DEEP

pytestmark = pytest.mark.gpu
DEV = "cuda"
GEOMS = [(4, 16, 3), (12, 16, 5), (8, 8, 5), (24, 8, 7), (16, 4, 11), (48, 4, 19)]   # (C, H=W, ragged B)


def make(c, fused):
    from normalizing_flow import AffineCoupling
    m = AffineCoupling(c, n_features=512, compute="bf16", fused=fused)
    sd = {k: T(synth_tensor("flows.3.affcoupling." + k, v.shape, 11, DEEP)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    return m, {"flows.3.affcoupling." + k: v for k, v in sd.items()}


@pytest.mark.parametrize("c,hw,b", GEOMS)
@torch.no_grad()
def test_fused_matches_unfused_and_oracle(c, hw, b):
    assert K.coupling_fused_supported(hw, hw, c, 512)
    fused, sd = make(c, True)
    plain, _ = make(c, False)
    x = T(synth_input(f"fx{c}", (b, c, hw, hw), 5)) * 0.6
    xd = x.to(DEV)
    l1 = torch.zeros(b, dtype=torch.float64, device=DEV)
    l2 = torch.zeros(b, dtype=torch.float64, device=DEV)
    K.start_trace()
    y1, _, _ = fused.transform(xd, l1, None)
    labels = [e["label"] for e in K.stop_trace()]
    assert any(lb.startswith("coupling_fused_kernel") for lb in labels), labels      # the fused kernel really ran
    y2, _, _ = plain.transform(xd, l2, None)
    assert torch.equal(y1[:, : c // 2], xd[:, : c // 2])
    assert_close(y1, y2, 6e-3, "fused vs unfused y")                 # summation order of two bf16 paths, element-wise: observed 3.2e-3
    assert_close(l1, l2, 3e-3, "fused vs unfused ldj")
    yo, lo = O.coupling_fwd(x, sd, "flows.3.affcoupling.net.", torch.zeros(b, dtype=torch.float64))
    assert_close(y1, yo, 2e-2, "fused vs fp32 oracle y")
    assert_close(l1, lo, 2e-2, "fused vs fp32 oracle ldj")
    i1, i2 = fused.invert(y1), plain.invert(y1)
    assert_close(i1, i2, 6e-3, "fused vs unfused inverse")
    assert relerr(i1, xd) < 2e-2                                                     # round trip at bf16 accuracy
    # determinism
    y1b, _, _ = fused.transform(xd, torch.zeros(b, dtype=torch.float64, device=DEV), None)
    assert torch.equal(y1, y1b)


@torch.no_grad()
def test_fused_ldj_const_and_none():
    fused, _ = make(12, True)
    x = (T(synth_input("fc", (4, 12, 16, 16), 6)) * 0.5).to(DEV)
    xh = K.to_nhwc(x).contiguous()
    base = torch.full((4,), 10.0, dtype=torch.float64, device=DEV)
    const = torch.tensor([1.5, -0.25], device=DEV)
    a, b = xh.clone(), xh.clone()
    l0 = torch.zeros(4, dtype=torch.float64, device=DEV)
    fused.transform_nhwc_(a, l0, None)
    l1 = base.clone()
    fused.transform_nhwc_(b, l1, const)
    assert torch.equal(a, b)
    assert_close(l1, l0 + 10.0 + 1.5 - 0.25, 1e-12, "ldj const")
    c2 = xh.clone()
    fused.transform_nhwc_(c2, None, None)                                             # no accumulator: still transforms
    assert torch.equal(a, c2)


@torch.no_grad()
def test_glow_m_golden_in_bf16_fused():
    """MNIST-shaped Glow (BASELINE configs[0] geometry) through the fused kernels vs the reference golden."""
    from normalizing_flow import Glow, GaussianPrior, calculate_loss
    from tests.util import load_golden, meta_of, synth_sd
    g = load_golden("glow_M.npz")
    m = meta_of(g)
    flow = Glow(in_channel=m["C"], L=m["L"], K=m["K"], compute="bf16")
    prior = GaussianPrior(2 ** (m["L"] + 1) * m["C"])
    flow.load_state_dict(synth_sd(m["flow_spec"], m["seed"]))
    prior.load_state_dict(synth_sd(m["prior_spec"], m["seed"]))
    x = T(g["x"]).to(DEV)
    ldj = torch.zeros(m["B"], dtype=torch.float64, device=DEV)
    logp = torch.zeros(m["B"], dtype=torch.float64, device=DEV)
    K.start_trace()
    zs, _, _ = flow.transform(x, ldj, logp)
    labels = [e["label"] for e in K.stop_trace()]
    n_level = sum(lb.startswith("level_fused") for lb in labels)
    assert n_level == m["L"], labels                                  # one persistent launch per Glow level
    assert not any(lb.startswith(("conv_kernel<", "coupling_kernel")) and "512" in lb for lb in labels)   # no unfused coupling nets
    logp += prior.compute_log_prob(zs[-1])
    assert_close(calculate_loss(ldj + logp, 32.0, 3072.0), g["bpd"], 2e-3, "bpd (bf16, golden weights are ill-conditioned)")
    assert_close(zs[-1], g[f"z{m['L'] - 1}"], 5e-2, "z_last")


@pytest.mark.parametrize("C,hw", [(16, 4), (48, 4), (8, 8), (24, 8)])
@pytest.mark.parametrize("inverse", [False, True])
@torch.no_grad()
def test_rowsplit_kernel_bit_identical(C, hw, inverse):
    """The 4x4 and 8x8 levels run row-split kernels (waves split the GEMM output rows of a 32- or 128-pixel workgroup,
    hidden tiles exchanged through LDS).  Same packed weights, same k order and rounding points as the
    wave-per-pixel-tile kernel: the two must agree bit for bit, for full and ragged batches."""
    import ctypes
    setter = K.lib().nfdp_debug_set_rowsplit
    setter.argtypes, setter.restype = [ctypes.c_int], None
    from normalizing_flow.glow import StepFlow
    torch.manual_seed(C + inverse)
    flow = StepFlow(C, 512, compute="bf16").to("cuda")
    sd = flow.state_dict()
    from tests.golden.synth import synth_tensor, DEEP
    import numpy as np
    flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor("flows.3." + k, v.shape, 5, DEEP))) for k, v in sd.items()})
    for B in (1, 2, 7, 64):
        x = torch.randn(B, hw, hw, C, device="cuda")
        outs = []
        for mode in (0, 1):
            setter(mode)
            xh = x.clone()
            ldj = torch.zeros(B, dtype=torch.float64, device="cuda")
            if inverse:
                flow.invert_nhwc_(xh)
            else:
                flow.transform_nhwc_(xh, ldj)
            outs.append((xh, ldj))
        setter(-1)
        assert torch.equal(outs[0][0], outs[1][0]), f"B={B}: state differs"
        assert torch.equal(outs[0][1], outs[1][1]), f"B={B}: ldj differs"
        assert torch.isfinite(outs[1][0]).all()


@pytest.mark.parametrize("C", [12, 4])
@pytest.mark.parametrize("inverse", [False, True])
@torch.no_grad()
def test_level0_16x16x32_form_against_32x32x16_form(C, inverse):
    """The 16x16 levels run their three GEMMs on v_mfma_f32_16x16x32_bf16 (image kind 2: 16-row x 32-deep fragments, hidden channels
    walked as chan16) - the same bf16 operands and fp32 accumulation as the 32x32x16 kernel, summed in a different order.  The two must
    agree to fp32-summation-order accuracy where no bf16 rounding of a hidden activation flips, and to bf16 accuracy overall; full and
    ragged batches; the log-det too."""
    import ctypes
    import numpy as np
    setter = K.lib().nfdp_debug_set_l0_k16
    setter.argtypes, setter.restype = [ctypes.c_int], None
    from normalizing_flow.glow import StepFlow
    from tests.golden.synth import synth_tensor, DEEP
    flow = StepFlow(C, 512, compute="bf16").to("cuda")
    flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor("flows.1." + k, v.shape, 11, DEEP))) for k, v in flow.state_dict().items()})
    torch.manual_seed(5 + inverse)
    try:
        for B in (1, 3, 64, 300):                             # 300: more images than CUs - some workgroups own two (TPW = 2)
            x = torch.randn(B, 16, 16, C, device="cuda")
            outs = []
            for k16 in (0, 1):
                setter(k16)
                assert K.coupling_fused_image_kind(16, 16, C) == 1 + k16
                xh = x.clone()
                ldj = torch.zeros(B, dtype=torch.float64, device="cuda")
                if inverse:
                    flow.invert_nhwc_(xh)
                else:
                    flow.transform_nhwc_(xh, ldj)
                outs.append((xh, ldj))
            assert torch.isfinite(outs[1][0]).all()
            assert_close(outs[1][0], outs[0][0], 6e-3, f"C={C} B={B} state, 16x16x32 vs 32x32x16")
            assert_close(outs[1][1], outs[0][1], 6e-3, f"C={C} B={B} ldj")
            assert not torch.equal(outs[0][0], x)
    finally:
        setter(-1)


@pytest.mark.parametrize("inverse", [False, True])
@torch.no_grad()
def test_level0_eight_wave_form_bit_identical(inverse):
    """The 16x16 level has a second form of its kernel: 8 waves x 32 pixels (two waves per SIMD at 256 registers) instead of 4 waves x
    64 pixels.  Same packed weights, same k order and rounding points: identical bits, full and ragged batches.  (It is not the default -
    measured 0-3 % slower, DESIGN.md 4.2 - and stays reachable through NFDP_L0_WAVES=8 for A/B runs.)"""
    import ctypes
    import numpy as np
    setter = K.lib().nfdp_debug_set_l0_waves
    setter.argtypes, setter.restype = [ctypes.c_int], None
    k16 = K.lib().nfdp_debug_set_l0_k16                      # both forms here are 32x32x16 kernels reading images of kind 1
    k16.argtypes, k16.restype = [ctypes.c_int], None
    k16(0)
    from normalizing_flow.glow import StepFlow
    from tests.golden.synth import synth_tensor, DEEP
    flow = StepFlow(12, 512, compute="bf16").to("cuda")
    flow.load_state_dict({k: torch.from_numpy(np.array(synth_tensor("flows.2." + k, v.shape, 7, DEEP))) for k, v in flow.state_dict().items()})
    torch.manual_seed(3 + inverse)
    try:
        for B in (1, 3, 64):
            x = torch.randn(B, 16, 16, 12, device="cuda")
            outs = []
            for nw in (4, 8):
                setter(nw)
                xh = x.clone()
                ldj = torch.zeros(B, dtype=torch.float64, device="cuda")
                if inverse:
                    flow.invert_nhwc_(xh)
                else:
                    flow.transform_nhwc_(xh, ldj)
                outs.append((xh, ldj))
            assert torch.equal(outs[0][0], outs[1][0]), f"B={B}: state differs"
            assert torch.equal(outs[0][1], outs[1][1]), f"B={B}: ldj differs"
            assert torch.isfinite(outs[1][0]).all()
    finally:
        setter(-1)
        k16(-1)


@pytest.mark.parametrize("C,hw,B", [(12, 16, 5), (12, 16, 300), (24, 8, 7), (48, 4, 9), (4, 16, 2), (4, 16, 257), (8, 8, 3), (16, 4, 5)])
@pytest.mark.parametrize("inverse", [False, True])
@torch.no_grad()
def test_level_launch_bit_identical_to_per_flow_launches(C, hw, B, inverse):
    """nfdp_level_fused_bf16 (K flows in one persistent launch, tile resident in LDS) against K single-flow launches:
    same arithmetic in the same order -> identical bits, log-det included; ragged batches."""
    import numpy as np
    from normalizing_flow.glow import StepFlow, _level_tables
    nfl = 5
    flows = []
    for f in range(nfl):
        m = StepFlow(C, 512, compute="bf16")
        m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(f"flows.{f}." + k, v.shape, 9, DEEP)))
                           for k, v in m.state_dict().items()})
        flows.append(m.to("cuda"))
    x = torch.randn(B, hw, hw, C, device="cuda") * 0.5
    ws = torch.stack([f.invconv2d._w() for f in flows])
    _, winvs = K.invconv_prepare(ws, want_inverse=True)
    a, b = x.clone(), x.clone()
    la = torch.zeros(B, dtype=torch.float64, device="cuda")
    lb = torch.zeros(B, dtype=torch.float64, device="cuda")
    if not inverse:
        for f in flows:
            f.transform_nhwc_(a, la, None, defer_logdet=True)
    else:
        for i in reversed(range(nfl)):
            flows[i].invert_nhwc_(a, winvs[i])
    packed, scales, biases = _level_tables(flows, b)
    wl = [winvs[i] for i in range(nfl)] if inverse else [f.invconv2d._w() for f in flows]
    K.level_fused(b, packed, scales, biases, wl, None if inverse else lb, inverse=inverse)
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), "state differs"
    assert torch.equal(la, lb), "log-det differs"


@torch.no_grad()
def test_empty_batch_is_a_no_op():
    """B = 0 (an empty shard of a ragged data-parallel split) must be accepted by the flow entry points and change nothing."""
    from normalizing_flow import Glow, GaussianPrior
    flow = Glow(in_channel=3, L=3, K=2, compute="bf16").to("cuda")
    for m in flow.modules():                                   # mark ActNorms initialised: nothing to initialise from
        if hasattr(m, "is_initialized"):
            m.is_initialized.fill_(1)
            m._init_cache = None
    x = torch.zeros(0, 3, 32, 32, device="cuda")
    ldj = torch.zeros(0, dtype=torch.float64, device="cuda")
    logp = torch.zeros(0, dtype=torch.float64, device="cuda")
    zs, ldj, logp = flow.transform(x, ldj, logp)
    assert [tuple(z.shape) for z in zs] == [(0, 6, 16, 16), (0, 12, 8, 8), (0, 48, 4, 4)]
    assert ldj.numel() == 0 and logp.numel() == 0
    xr = flow.invert(zs)
    assert tuple(xr.shape) == (0, 3, 32, 32)


@pytest.mark.parametrize("c,hw,b", [(12, 64, 2), (12, 20, 3), (24, 32, 2), (48, 16, 3), (4, 32, 2), (8, 24, 2), (16, 12, 8)])
@torch.no_grad()
def test_tiled_coupling_network_any_image_size(c, hw, b):
    """The coupling network for images larger than the fused flow-step kernels' (im2col -> per-pixel GEMM chain with the hidden
    activations on chip -> tap shift-sum) against the three bf16 convs one by one, incl. pixel counts that are not a multiple
    of the 128/256-pixel workgroup tile, and the coupling forward/inverse built on it."""
    from normalizing_flow.transforms import AffineCoupling
    cp = AffineCoupling(in_channels=c, n_features=512, compute="bf16").to(DEV)
    g = torch.Generator().manual_seed(c * 7 + hw)
    for name, prm in cp.net.named_parameters():
        prm.data.copy_((torch.randn(prm.shape, generator=g) * (0.05 if prm.dim() == 4 else 0.2)).to(DEV))
    for m in (cp.net[0].actnorm, cp.net[2].actnorm):
        m.is_initialized.fill_(1)
    x = (torch.randn(b, hw, hw, c, generator=g) * 0.7).to(DEV)
    h_ref = cp.net.forward_nhwc(x[..., : c // 2], compute="bf16")
    h = cp._net_nhwc(x)
    assert h.shape == h_ref.shape
    scale = float(h_ref.abs().max())
    assert float((h - h_ref).abs().max()) < 2e-2 * scale, (float((h - h_ref).abs().max()), scale)
    assert float((h - h_ref).pow(2).mean().sqrt()) < 3e-3 * scale
    y = x.clone()
    ldj = torch.zeros(b, dtype=torch.float64, device=DEV)
    cp.transform_nhwc_(y, ldj)
    assert torch.isfinite(y).all() and torch.isfinite(ldj).all()
    xr = cp.invert_nhwc_(y.clone())
    assert float((xr - x).abs().max()) < 1e-4 * float(x.abs().max())       # same NN(x_a) both ways: exact up to the 1e-6 quirk


# ----------------------------------------------------------------------------- exact fp32, hidden activations on chip
F32_GEOMS = [(4, 16, 16, 3), (12, 16, 16, 5), (8, 8, 8, 5), (24, 8, 8, 7), (16, 4, 4, 11), (48, 4, 4, 19), (12, 6, 10, 3), (24, 32, 32, 1)]


@pytest.mark.parametrize("c,hh,ww,b", F32_GEOMS)
@torch.no_grad()
def test_f32_onchip_coupling_network_against_oracle(c, hh, ww, b):
    """nfdp_coupling_mlp_f32 (3x3 gather + three exact-fp32 MFMA GEMMs, hidden activations in registers) + tap shift-sum = NN(x_a) of
    utils.py:83-89 in fp32: against the CPU oracle and the conv-by-conv fp32 HIP path, every supported channel count, square and
    non-square images, pixel counts that are not a multiple of the 128-pixel workgroup; then the whole coupling layer forward /
    inverse through AffineCoupling(compute="f32") on a channel-strided view (x_ld = C)."""
    from normalizing_flow import AffineCoupling
    m = AffineCoupling(c, n_features=512, compute="f32")
    sd = {k: T(synth_tensor("flows.3.affcoupling." + k, v.shape, 11, DEEP)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    osd = {"flows.3.affcoupling." + k: v for k, v in sd.items()}
    x = T(synth_input(f"f32x{c}_{hh}", (b, c, hh, ww), 5)) * 0.6
    xh = K.to_nhwc(x.to(DEV)).contiguous()
    assert K.coupling_mlp_f32_supported(c, 512)
    K.start_trace()
    h_fused = m._net_nhwc(xh)
    labels = [e["label"] for e in K.stop_trace()]
    assert any(lb.startswith("coupling_mlp_f32") for lb in labels), labels          # the on-chip kernel really ran
    h_ref = O.coupling_net(x[:, : c // 2], osd, "flows.3.affcoupling.net.")
    assert_close(K.as_nchw(h_fused), h_ref, 2e-5, "on-chip fp32 network vs oracle")
    zc = m.net[-1]
    for variant, what in ((1, "one wave per tile"), (2, "four waves per tile, rows split")):     # both kernels, whatever the auto choice was
        hv = K.coupling_net_tiled_f32(xh[..., : c // 2], m.net.mlp_f32_packed(), c, zc.bias, zc.logs.view(-1), zc.logscale_factor,
                                      variant=variant)
        assert_close(K.as_nchw(hv), h_ref, 2e-5, f"on-chip fp32 network ({what}) vs oracle")
    K.set_f32_fused(False)
    try:
        h_plain = m._net_nhwc(xh)
    finally:
        K.set_f32_fused(True)
    assert_close(h_fused, h_plain, 2e-5, "on-chip fp32 network vs conv-by-conv fp32")
    ldj = torch.zeros(b, dtype=torch.float64, device=DEV)
    y, ldj, _ = m.transform(x.to(DEV), ldj, None)
    yo, lo = O.coupling_fwd(x, osd, "flows.3.affcoupling.net.", torch.zeros(b, dtype=torch.float64))
    assert_close(y, yo, 1e-5, "coupling forward")
    assert_close(ldj, lo, 1e-6, "coupling log-det")
    assert_close(m.invert(y), O.coupling_inv(yo, osd, "flows.3.affcoupling.net."), 1e-5, "coupling inverse")


# ---- round 4: exact-fp32 mode with split products (three bf16 MFMAs per fp32 product)
X3_GEOMS = [(4, 16, 16, 3), (12, 16, 16, 5), (8, 8, 8, 5), (24, 8, 8, 7), (16, 4, 4, 11), (12, 6, 10, 3), (24, 32, 32, 1), (12, 64, 64, 1)]


@pytest.mark.parametrize("c,hh,ww,b", X3_GEOMS)
@torch.no_grad()
def test_split_product_coupling_network_against_fp64(c, hh, ww, b):
    """nfdp_coupling_mlp_gather_bf16x3 (K.set_f32_coupling("bf16x3")): NN(x_a) of utils.py:83-89 with every product formed from two-piece
    bf16 splits of both operands, against the oracle evaluated in fp64 - 1e-4 of the tensor's scale is the bar, measured 1-2e-5
    (bf16 operands: 5e-3; fp32 MFMA: 1e-6); square / non-square images, ragged pixel counts; then the coupling layer forward and
    inverse through AffineCoupling(compute="f32"); and the kernel is deterministic."""
    from normalizing_flow import AffineCoupling
    m = AffineCoupling(c, n_features=512, compute="f32")
    sd = {k: T(synth_tensor("flows.3.affcoupling." + k, v.shape, 11, DEEP)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    osd64 = {"flows.3.affcoupling." + k: v.double() for k, v in sd.items()}
    x = T(synth_input(f"f32x{c}_{hh}", (b, c, hh, ww), 5)) * 0.6
    xh = K.to_nhwc(x.to(DEV)).contiguous()
    zc = m.net[-1]
    h_ref = O.coupling_net(x[:, : c // 2].double(), osd64, "flows.3.affcoupling.net.")
    K.set_f32_coupling("bf16x3")
    try:
        assert K.coupling_mlp_x3_supported(xh[..., : c // 2], c, 512)
        K.start_trace()
        t = m._net_taps_nhwc(xh)
        labels = [e["label"] for e in K.stop_trace()]
        assert t is not None and any(lb.startswith("coupling_mlp_x3") for lb in labels), labels
        h3 = K.coupling_net_tiled(xh[..., : c // 2], m.net.fused_packed(1), c, zc.bias, zc.logs.view(-1), zc.logscale_factor,
                                  packed_lo=m.net.fused_packed(3))
        h3b = K.coupling_net_tiled(xh[..., : c // 2], m.net.fused_packed(1), c, zc.bias, zc.logs.view(-1), zc.logscale_factor,
                                   packed_lo=m.net.fused_packed(3))
        assert torch.equal(h3, h3b)
        e3 = relerr(K.as_nchw(h3), h_ref)
        record(f"x3_network_relerr_c{c}_h{hh}", e3)
        assert e3 < 1e-4, e3
        h16 = K.coupling_net_tiled(xh[..., : c // 2], m.net.fused_packed(1), c, zc.bias, zc.logs.view(-1), zc.logscale_factor)
        assert e3 < 0.05 * relerr(K.as_nchw(h16), h_ref)                     # two orders below bf16 operands
        ldj = torch.zeros(b, dtype=torch.float64, device=DEV)
        y, ldj, _ = m.transform(x.to(DEV), ldj, None)
        yo, lo = O.coupling_fwd(x.double(), osd64, "flows.3.affcoupling.net.", torch.zeros(b, dtype=torch.float64))
        assert_close(y, yo, 1e-4, "coupling forward (split products)")
        assert_close(ldj, lo, 1e-5, "coupling log-det (split products)")
        assert_close(m.invert(y), O.coupling_inv(yo, osd64, "flows.3.affcoupling.net."), 1e-4, "coupling inverse (split products)")
    finally:
        K.set_f32_coupling("mfma_f32")


# ---- round 4: the wide levels of a 5-level model (8x8x96 with the flow head, 4x4x192 coupling only) and the one-pass tap + coupling kernel
WIDE = [(96, 8, 5), (192, 4, 5), (192, 4, 2)]          # (C, H=W, ragged B: 4x4 workgroups own two images)


@pytest.fixture(params=[True, False], ids=["split", "one-workgroup"])
def wide_split(request):
    """The wide levels' two kernel families: a pixel tile on four workgroups and three launches per flow (fused_split.inc, the default) or
    one workgroup per tile (coupling_fused_wide_kernel)."""
    K.set_wide_levels_split(request.param)
    yield request.param
    K.set_wide_levels_split(True)


@pytest.mark.parametrize("c,hw,b", WIDE)
@torch.no_grad()
def test_wide_level_kernels_match_the_conv_by_conv_path_and_the_oracle(c, hw, b, wide_split):
    """coupling_fused_wide_kernel (inference only: the 3x3 output conv as one accumulation over (tap, hidden channel), the hidden tile
    read at the shifted pixel) and the split launches (split_gemm1 / split_gemm23 / split_coupling: the conv's sum by hidden block) against
    the same layer conv by conv in bf16 mode (same operands, another fp32 association of the tap sums) and against the fp32 oracle;
    forward, inverse, round trip, determinism."""
    assert not K.coupling_fused_supported(hw, hw, c, 512) and K.coupling_fused_infer_supported(hw, hw, c, 512)
    assert K.coupling_fused_has_head(hw, hw, c) == (c == 96)
    fused, sd = make(c, True)
    plain, _ = make(c, False)
    x = T(synth_input(f"wx{c}", (b, c, hw, hw), 5)) * 0.6
    xd = x.to(DEV)
    l1 = torch.zeros(b, dtype=torch.float64, device=DEV)
    l2 = torch.zeros(b, dtype=torch.float64, device=DEV)
    K.start_trace()
    y1, _, _ = fused.transform(xd, l1, None)
    labels = [e["label"] for e in K.stop_trace()]
    assert any(lb.startswith("coupling_split" if wide_split else "coupling_fused_kernel") for lb in labels), labels
    y2, _, _ = plain.transform(xd, l2, None)
    assert torch.equal(y1[:, : c // 2], xd[:, : c // 2])
    assert_close(y1, y2, 1e-2, "wide vs conv-by-conv y")
    assert_close(l1, l2, 3e-3, "wide vs conv-by-conv ldj")
    yo, lo = O.coupling_fwd(x, sd, "flows.3.affcoupling.net.", torch.zeros(b, dtype=torch.float64))
    assert_close(y1, yo, 3e-2, "wide vs fp32 oracle y")
    assert_close(l1, lo, 2e-2, "wide vs fp32 oracle ldj")
    i1, i2 = fused.invert(y1), plain.invert(y1)
    assert_close(i1, i2, 1e-2, "wide vs conv-by-conv inverse")
    assert relerr(i1, xd) < 2e-2
    y1b, _, _ = fused.transform(xd, torch.zeros(b, dtype=torch.float64, device=DEV), None)
    assert torch.equal(y1, y1b)


@pytest.mark.parametrize("c,hw", [(96, 8), (192, 4)])
@torch.no_grad()
def test_wide_level_launch_of_several_flows_matches_flow_by_flow(wide_split, c, hw):
    """A wide level as ONE call of several flows with the flow heads inside - the 8x8x96 level on either kernel family (one persistent
    launch: weight ring continuous across flows, parameters fetched a flow ahead; or 2 n + 1 split launches: a flow's partial sums and
    coupling inside the next flow's first launch), the 4x4x192 level on the split launches (head_apply_wide) - against the same flows conv
    by conv: forward, and the inverse of the unfused output back to the input."""
    import numpy as np
    from normalizing_flow import StepFlow
    from normalizing_flow.glow import _run_flows_fwd, _run_flows_inv
    if c == 192 and not wide_split:
        pytest.skip("the one-workgroup kernel of the 4x4x192 level is the coupling layer alone: no level call")
    flows = []
    for f in range(3):
        m = StepFlow(c, 512, compute="bf16")
        m.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(f"flows.{f}." + k, v.shape, 1, DEEP))) for k, v in m.state_dict().items()})
        flows.append(m.to(DEV))
    x = torch.rand(5, hw, hw, c, device=DEV) - 0.5
    out = {}
    for fused in (True, False):
        for m in flows:
            m.affcoupling.fused = fused
        K.start_trace()
        ldj = torch.zeros(5, dtype=torch.float64, device=DEV)
        y = _run_flows_fwd(flows, x.clone(), ldj, None, defer_logdet=True)
        n_launch = len([e for e in K.stop_trace() if e["label"].startswith(("level_fused", "level_split", "coupling_fused_kernel"))])
        out[fused] = (y, ldj, n_launch)
    assert out[True][2] == 1 and out[False][2] == 0, (out[True][2], out[False][2])      # one level launch vs none
    assert_close(out[True][0], out[False][0], 1e-2, "level launch vs flow by flow")
    assert_close(out[True][1], out[False][1], 3e-3, "level launch vs flow by flow: ldj")
    for m in flows:
        m.affcoupling.fused = True
    K.start_trace()
    xi = _run_flows_inv(flows, out[False][0].clone())
    assert len([e for e in K.stop_trace() if e["label"].startswith(("level_fused", "level_split"))]) == 1
    assert float((xi - x).abs().max()) < 1e-2


@pytest.mark.parametrize("c,h,w,b", [(12, 64, 64, 3), (24, 32, 32, 2), (48, 16, 16, 3), (96, 8, 8, 5), (192, 4, 4, 7), (12, 20, 12, 2)])
@torch.no_grad()
def test_tap_coupling_is_bit_identical_to_shift_sum_then_coupling(c, h, w, b):
    """nfdp_tap_coupling_f32 (tap shift-sum + ZeroConv2d epilogue + coupling + log-det in one pass) against nfdp_tap_shift_sum_f32
    followed by nfdp_coupling_fwd_f32 / _inv_f32: same expressions, same tap order -> bit-identical x and ldj, chunked and unchunked
    geometries, non-square images."""
    g = torch.Generator().manual_seed(c * 7 + h)
    t = (torch.randn(b, h, w, 9 * c, generator=g) * 0.3).to(DEV)
    x0 = (torch.randn(b, h, w, c, generator=g)).to(DEV)
    add, mul = (torch.randn(c, generator=g) * 0.1).to(DEV), (torch.randn(c, generator=g) * 0.05).to(DEV)
    const = torch.tensor([0.75, -1.25], device=DEV)
    hh = torch.empty(b, h, w, c, device=DEV)
    K._call("nfdp_tap_shift_sum_f32", K._ptr(t), K._ptr(hh), b, h, w, c, K._ptr(add), K._ptr(mul), 3.0, 0, K._stream())
    for inverse in (False, True):
        xa, xb = x0.clone(), x0.clone()
        la = torch.full((b,), 2.0, dtype=torch.float64, device=DEV)
        lb = la.clone()
        if inverse:
            K.coupling_inv(xa, hh)
        else:
            K.coupling_fwd(xa, hh, la, const)
        K.tap_coupling(xb, t, add, mul, 3.0, lb, const, inverse=inverse)
        assert torch.equal(xa, xb), (inverse, float((xa - xb).abs().max()))
        if not inverse:
            assert float((la - lb).abs().max()) <= 1e-9 * float(la.abs().max()), (la, lb)      # (chunk counts differ: fp64 sums of the same terms)


@pytest.mark.parametrize("c,h,w,b", [(12, 64, 64, 3), (24, 32, 32, 2), (48, 16, 16, 3), (12, 20, 12, 2), (4, 14, 14, 5), (8, 7, 7, 4), (16, 6, 10, 3),
                                     (32, 8, 8, 2)])
@torch.no_grad()
def test_tap_coupling_with_the_adjoining_head_is_bit_identical_to_two_launches(c, h, w, b):
    """nfdp_tap_coupling_head_f32: forward = tap + coupling pass, then the NEXT flow's ActNorm + 1x1 conv on the finished pixel;
    inverse = the pass, then THIS flow's inverse head - against nfdp_tap_coupling_f32 followed by the head launches
    (nfdp_flow_head_fwd_f32 / nfdp_invconv_actnorm_inv_f32): same expressions in the same order -> bit-identical flow state and log-det
    constants; ldj to fp64 summation order.  Ragged pixel groups (85 / 42 pixels per workgroup pass), chunked and unchunked images."""
    assert K.tap_coupling_head_supported(c)
    g = torch.Generator().manual_seed(c * 11 + h)
    t = (torch.randn(b, h, w, 9 * c, generator=g) * 0.3).to(DEV)
    x0 = (torch.randn(b, h, w, c, generator=g)).to(DEV)
    add, mul = (torch.randn(c, generator=g) * 0.1).to(DEV), (torch.randn(c, generator=g) * 0.05).to(DEV)
    hs, hb = (torch.randn(c, generator=g) * 0.2).to(DEV), (torch.randn(c, generator=g) * 0.3).to(DEV)
    wm = (torch.linalg.qr(torch.randn(c, c, generator=g))[0] + 0.05 * torch.randn(c, c, generator=g)).contiguous().to(DEV)
    lad = torch.tensor([0.3125], device=DEV)
    const = torch.tensor([0.75, -1.25], device=DEV)
    for inverse in (False, True):
        xa, xb = x0.clone(), x0.clone()
        la = torch.full((b,), 2.0, dtype=torch.float64, device=DEV)
        lb = la.clone()
        K.tap_coupling(xa, t, add, mul, 3.0, la, const, inverse=inverse)
        if inverse:
            K.invconv_actnorm_inv(xa, hs, hb, wm, out=xa)
            K.tap_coupling_head(xb, t, add, mul, 3.0, None, None, True, hs, hb, wm.reshape(-1))
        else:
            _, ca = K.flow_head_fwd(xa, hs, hb, wm, lad, out=xa)
            cb = K.tap_coupling_head(xb, t, add, mul, 3.0, lb, const, False, hs, hb, wm, lad, want_const=True)
            assert torch.equal(ca, cb), (ca, cb)
            assert float((la - lb).abs().max()) <= 1e-9 * float(la.abs().max()), (la, lb)
            # deferred log-det sums: two launches leave their chunk sums behind, one finish adds both - as two immediate launches do
            xc, xd = x0.clone(), x0.clone()
            lc, ld = torch.full((b,), 2.0, dtype=torch.float64, device=DEV), torch.full((b,), 2.0, dtype=torch.float64, device=DEV)
            pend = []
            for xx, ll, dd in ((xc, lc, None), (xd, ld, pend)):
                K.tap_coupling_head(xx, t, add, mul, 3.0, ll, const, False, hs, hb, wm, lad, defer=dd)
                K.tap_coupling_head(xx, t, add, mul, 3.0, ll, None, False, hs, hb, wm, lad, defer=dd)
            if pend:                                                                  # (chunked geometries defer; one-chunk launches add themselves)
                assert len(pend) == 2 and torch.equal(ld, torch.full_like(ld, 2.0))
            K.ldj_finish_level(ld, pend)
            assert torch.equal(xc, xd) and torch.equal(lc, ld), float((lc - ld).abs().max())
        assert torch.equal(xa, xb), (inverse, float((xa - xb).abs().max()))


@torch.no_grad()
def test_level_loop_with_fused_heads_matches_flow_by_flow():
    """glow._run_flows_fwd / _run_flows_inv on a tiled level (16x16x48, the third level of a 128x128 model): with the adjoining heads
    inside the tap + coupling passes (K.TAP_HEAD_FUSION, default) and without - bit-identical flow state, ldj to fp64 summation order,
    and one launch less per flow."""
    from normalizing_flow.glow import StepFlow, _run_flows_fwd, _run_flows_inv
    c, hw, b, nf = 48, 16, 6, 3
    flows = []
    for i in range(nf):
        f = StepFlow(in_channels=c, compute="bf16")
        f.load_state_dict({k: T(synth_tensor(f"blocks.2.flows.{i}." + k, v.shape, 3, DEEP)) for k, v in f.state_dict().items()}, strict=True)
        flows.append(f.to(DEV))
    x0 = K.to_nhwc(T(synth_input("lvl48", (b, c, hw, hw), 9)).to(DEV) * 0.6).contiguous()
    outs = {}
    for fused in (True, False):
        K.TAP_HEAD_FUSION = fused
        try:
            xa = x0.clone()
            la = torch.zeros(b, dtype=torch.float64, device=DEV)
            K.start_trace()
            _run_flows_fwd(flows, xa, la, None, defer_logdet=True)
            n_fwd = len([e for e in K.stop_trace() if not e["label"].startswith(("nfdp_coupling_fused_pack", "nfdp_ldj_finish_level"))])
            ya = xa.clone()
            _run_flows_inv(flows, ya)
            outs[fused] = (xa, la, ya, n_fwd)
        finally:
            K.TAP_HEAD_FUSION = True
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][2], outs[False][2])
    # (ldj: the fused passes defer their chunk sums to ONE finish launch per level that adds them in flow order; the two-launch path
    #  uses other chunk counts: fp64 sums of the same terms)
    assert float((outs[True][1] - outs[False][1]).abs().max()) <= 1e-9 * float(outs[False][1].abs().max())
    assert outs[True][3] == outs[False][3] - (nf - 1), (outs[True][3], outs[False][3])
    assert relerr(outs[True][2], x0) < 2e-2                                            # round trip at bf16 accuracy


ROWS_GEOMS = [(48, 16, 16, 3), (48, 4, 4, 19), (24, 8, 8, 7), (24, 6, 10, 3), (12, 16, 16, 5), (12, 7, 5, 3), (48, 3, 5, 1)]


@pytest.mark.parametrize("c,hh,ww,b", ROWS_GEOMS)
@torch.no_grad()
def test_row_split_tiled_network_is_bit_identical_to_the_128_pixel_form(c, hh, ww, b):
    """coupling_mlp_rows_kernel (32-pixel workgroups, the waves split the output rows of every GEMM, hidden blocks through LDS) against
    coupling_mlp_kernel (a wave walks its 32 pixels through all weights): same images, same bf16 operands, same k order in every
    accumulator -> bit-identical tap contributions; ragged pixel counts (tail workgroup), images smaller than a tile, non-square images;
    and against the oracle at the bf16 tolerance."""
    from normalizing_flow import AffineCoupling
    m = AffineCoupling(c, n_features=512, compute="bf16")
    sd = {k: T(synth_tensor("flows.5.affcoupling." + k, v.shape, 13, DEEP)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    x = T(synth_input(f"rows{c}_{hh}", (b, c, hh, ww), 5)) * 0.6
    xh = K.to_nhwc(x.to(DEV)).contiguous()
    zc = m.net[-1]
    outs = {}
    try:
        for mode in ("never", "always"):
            K.set_mlp_rows(mode)
            outs[mode] = K.coupling_net_tiled(xh[..., : c // 2], m.net.fused_packed(1), c, zc.bias, zc.logs.view(-1), zc.logscale_factor,
                                              return_taps=True).clone()
        K.set_mlp_rows("always")
        again = K.coupling_net_tiled(xh[..., : c // 2], m.net.fused_packed(1), c, zc.bias, zc.logs.view(-1), zc.logscale_factor, return_taps=True)
        h_rows = K.coupling_net_tiled(xh[..., : c // 2], m.net.fused_packed(1), c, zc.bias, zc.logs.view(-1), zc.logscale_factor)
    finally:
        K.set_mlp_rows("auto")
    assert torch.equal(outs["never"], outs["always"]), float((outs["never"] - outs["always"]).abs().max())
    assert torch.equal(outs["always"], again)
    osd = {"flows.5.affcoupling." + k: v for k, v in sd.items()}
    assert relerr(K.as_nchw(h_rows), O.coupling_net(x[:, : c // 2], osd, "flows.5.affcoupling.net.")) < 1e-2      # bf16 operands


@pytest.mark.parametrize("c,hh,ww,b", [(48, 4, 4, 19), (48, 16, 16, 2), (24, 8, 8, 7), (12, 6, 10, 3)])
@torch.no_grad()
def test_row_split_split_product_network_against_fp64(c, hh, ww, b):
    """The row-split kernel in its split-product form (NP = 2: high / low images, three bf16 MFMAs per fp32 product; the only
    split-product form at C = 48) against the oracle in fp64: 1e-4 of the tensor's scale is the bar, measured below 1e-5."""
    from normalizing_flow import AffineCoupling
    m = AffineCoupling(c, n_features=512, compute="f32")
    sd = {k: T(synth_tensor("flows.3.affcoupling." + k, v.shape, 11, DEEP)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    osd64 = {"flows.3.affcoupling." + k: v.double() for k, v in sd.items()}
    x = T(synth_input(f"f32x{c}_{hh}", (b, c, hh, ww), 5)) * 0.6
    xh = K.to_nhwc(x.to(DEV)).contiguous()
    zc = m.net[-1]
    h_ref = O.coupling_net(x[:, : c // 2].double(), osd64, "flows.3.affcoupling.net.")
    K.set_f32_coupling("bf16x3")
    K.set_mlp_rows("always")
    try:
        assert K.coupling_mlp_x3_supported(xh[..., : c // 2], c, 512)
        h3 = K.coupling_net_tiled(xh[..., : c // 2], m.net.fused_packed(1), c, zc.bias, zc.logs.view(-1), zc.logscale_factor,
                                  packed_lo=m.net.fused_packed(3))
        e3 = relerr(K.as_nchw(h3), h_ref)
        record(f"x3_rows_network_relerr_c{c}_h{hh}", e3)
        assert e3 < 1e-4, e3
        ldj = torch.zeros(b, dtype=torch.float64, device=DEV)
        y, ldj, _ = m.transform(x.to(DEV), ldj, None)
        yo, lo = O.coupling_fwd(x.double(), osd64, "flows.3.affcoupling.net.", torch.zeros(b, dtype=torch.float64))
        assert_close(y, yo, 1e-4, "coupling forward (row-split split products)")
        assert_close(ldj, lo, 1e-5, "coupling log-det (row-split split products)")
    finally:
        K.set_mlp_rows("auto")
        K.set_f32_coupling("mfma_f32")


@pytest.mark.parametrize("c,hh,ww,b", [(12, 64, 64, 5), (12, 16, 16, 3), (4, 14, 14, 5), (12, 6, 10, 3), (4, 3, 5, 1), (12, 32, 32, 17)])
@torch.no_grad()
def test_k16_tiled_network_matches_the_32x32_form_and_the_oracle(c, hh, ww, b):
    """coupling_mlp16_kernel (the tiled network on v_mfma_f32_16x16x32_bf16, images of kinds 2 / 4) against coupling_mlp_kernel /
    coupling_mlp_x3_kernel (kinds 1 / 3): same bf16 operands / pieces, another grouping of the fp32 accumulation (32-deep k-steps) -
    bf16: within a few bf16 roundings of hidden activations (2e-3 of the tensor's scale), split products: 1e-5 between the two; the split-product form against the oracle in fp64 (1e-4 is the bar); ragged pixel
    counts, images smaller than a tile; determinism; and the auto choice (nfdp_coupling_mlp_image_kind)."""
    from normalizing_flow import AffineCoupling
    m = AffineCoupling(c, n_features=512, compute="f32")
    sd = {k: T(synth_tensor("flows.7.affcoupling." + k, v.shape, 17, DEEP)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    osd64 = {"flows.7.affcoupling." + k: v.double() for k, v in sd.items()}
    x = T(synth_input(f"k16_{c}_{hh}", (b, c, hh, ww), 5)) * 0.6
    xh = K.to_nhwc(x.to(DEV)).contiguous()
    xa = xh[..., : c // 2]
    zc = m.net[-1]
    args = (c, zc.bias, zc.logs.view(-1), zc.logscale_factor)
    K.set_mlp_rows("never")
    try:
        t32 = K.coupling_net_tiled(xa, m.net.fused_packed(1), *args, return_taps=True)
        t16 = K.coupling_net_tiled(xa, m.net.fused_packed(2), *args, return_taps=True, kind=2)
        t16b = K.coupling_net_tiled(xa, m.net.fused_packed(2), *args, return_taps=True, kind=2)
        assert torch.equal(t16, t16b)
        assert relerr(t16, t32) < 2e-3, relerr(t16, t32)      # (bf16 operands: an fp32 sum an ulp apart can round a hidden activation the other way)
        x32 = K.coupling_net_tiled(xa, m.net.fused_packed(1), *args, packed_lo=m.net.fused_packed(3))
        x16 = K.coupling_net_tiled(xa, m.net.fused_packed(2), *args, packed_lo=m.net.fused_packed(4), kind=2)
        assert relerr(x16, x32) < 1e-5, relerr(x16, x32)
        h_ref = O.coupling_net(x[:, : c // 2].double(), osd64, "flows.7.affcoupling.net.")
        e = relerr(K.as_nchw(x16), h_ref)
        record(f"x3_k16_network_relerr_c{c}_h{hh}", e)
        assert e < 1e-4, e
    finally:
        K.set_mlp_rows("auto")
    npix = b * hh * ww
    want = 2 if (c == 4 or npix > 16384) else 1                       # C = 12 up to 16384 pixels: the row-split form (kind-1 images)
    assert K.coupling_mlp_image_kind(xa, c) == want and K.coupling_mlp_image_kind(xa, c, split=True) == want
    K.set_mlp_k16(False)
    try:
        assert K.coupling_mlp_image_kind(xa, c) == 1
    finally:
        K.set_mlp_k16(True)
