"""GPU parity, per op: HIP kernels (through the C ABI and the reference-shaped classes) vs the golden vectors
captured from the reference and vs the CPU oracle on seeded inputs.

Tolerances: index reshuffles bit-exact; fp32 mode 1e-4 relative (BASELINE.json north_star; observed ~1e-6);
bf16 mode is compared at a separately stated, looser tolerance.
"""
import numpy as np
import pytest
import torch

import nfdp_hip as K
from oracle import nf_oracle as O
from tests.util import load_golden, synth_sd, T, assert_close, relerr
from tests.golden.synth import synth_tensor, synth_input

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    return load_golden("ops.npz")


def cu(a):
    return T(a).to(DEV)


def z64(b):
    return torch.zeros(b, dtype=torch.float64, device=DEV)


def test_layout_roundtrip_bit_exact():
    x = torch.randn(3, 5, 6, 4, device=DEV)
    xh = K.to_nhwc(x)
    assert xh.shape == (3, 6, 4, 5) and xh.is_contiguous()
    assert torch.equal(xh, x.permute(0, 2, 3, 1))
    assert torch.equal(K.to_nchw_contiguous(xh), x)
    assert K.to_nhwc(K.as_nchw(xh)).data_ptr() == xh.data_ptr()       # zero-copy when already channels_last


def test_squeeze_bit_exact(ops):
    from normalizing_flow import Squeeze
    sq = Squeeze()
    x = cu(ops["squeeze_x"])
    y, _, _ = sq.transform(x, None, None)
    assert torch.equal(y.cpu(), T(ops["squeeze_y"]))
    assert torch.equal(sq.invert(y).cpu(), T(ops["squeeze_x"]))
    big = torch.randn(512, 3, 32, 32, device=DEV)                        # BASELINE size: oracle permutation, exact
    yb, _, _ = sq.transform(big, None, None)
    assert torch.equal(yb.cpu(), O.squeeze(big.cpu()))
    assert torch.equal(sq.invert(yb), big)
    with pytest.raises(ValueError):
        sq.transform(torch.zeros(1, 3, 5, 4, device=DEV), None, None)


@pytest.mark.parametrize("c", [4, 12, 48])
def test_actnorm(ops, c):
    from normalizing_flow import ActNorm
    m = ActNorm(c)
    x = cu(ops[f"actnorm{c}_x"])
    ldj = z64(4)
    y, ldj2, _ = m.transform(x, ldj, None)
    assert ldj2 is ldj and int(m.is_initialized) == 1
    assert_close(m.scale, ops[f"actnorm{c}_scale"], TOL, "scale")
    assert_close(m.bias, ops[f"actnorm{c}_bias"], TOL, "bias")
    assert_close(y, ops[f"actnorm{c}_y"], TOL, "y")
    assert_close(ldj, ops[f"actnorm{c}_ldj"], TOL, "ldj")
    assert_close(m.invert(y), ops[f"actnorm{c}_inv"], TOL, "inv")


def test_actnorm_reference_unittest():
    """tests/transformations.py:10-42 of the reference, 1:1 (fp32 accumulators, EPS=1e-3)."""
    from normalizing_flow import ActNorm
    x = torch.randn(32, 3, 28, 28, device=DEV)
    f = ActNorm(in_channels=3)
    logdet = torch.zeros(32, device=DEV)
    y, logdet, _ = f.transform(x, logdet, torch.zeros(32, device=DEV))
    assert (f.invert(y) - x).norm() < 1e-3
    assert y.mean(dim=(0, 2, 3)).norm() < 1e-3
    assert (y.var(dim=(0, 2, 3)) - 1).norm() < 1e-3
    assert logdet.dtype == torch.float32 and torch.isfinite(logdet).all()


def test_invconv_reference_unittest():
    """tests/transformations.py:45-65 of the reference, 1:1: `TestInvConv2d` - x ~ N(0, 1) of shape [32, 3, 28, 28], a freshly constructed
    layer (random orthogonal weight, transforms.py:112-114), fp32 accumulators, ||invert(transform(x)) - x||_2 < EPS = 1e-3."""
    from normalizing_flow import InvConv2d
    x = torch.randn(32, 3, 28, 28, device=DEV)
    f = InvConv2d(in_channels=3).to(DEV)
    logdet, logp = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
    y, logdet, logp = f.transform(x, logdet, logp)
    inv_y = f.invert(y)
    assert x.size() == y.size()
    assert (inv_y - x).norm() < 1e-3
    assert logdet.dtype == torch.float32 and torch.isfinite(logdet).all()


def test_affine_coupling_reference_unittest():
    """tests/transformations.py:68-86 of the reference, 1:1: `TestAffCoupling` - x ~ N(0, 1) of shape [32, 4, 28, 28], a freshly constructed
    AffineCoupling(4) (its ZeroConv2d is zero, utils.py:17-44: log_s = t = 0, s = sigmoid(2)), fp32 accumulators,
    ||invert(transform(x)) - x||_2 < EPS = 1e-3.  28x28 is outside the fused kernels' table: the per-op path."""
    from normalizing_flow import AffineCoupling
    x = torch.randn(32, 4, 28, 28, device=DEV)
    f = AffineCoupling(4).to(DEV)
    logdet, logp = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
    y, logdet, logp = f.transform(x, logdet, logp)
    inv_y = f.invert(y)
    assert x.size() == y.size() == inv_y.size()
    assert (inv_y - x).norm() < 1e-3
    # what the zero-initialised network implies (transforms.py:179-185): x_a passes, x_b is scaled by sigmoid(2), log-det = 2 * 28 * 28 * log(s + 1e-6)
    s = 1.0 / (1.0 + np.exp(-2.0))
    assert torch.equal(y[:, :2], x[:, :2])
    assert (y[:, 2:] - x[:, 2:] * s).abs().max() < 1e-6
    assert (logdet - 2 * 28 * 28 * np.log(s + 1e-6)).abs().max() < 1e-2


@pytest.mark.parametrize("c", [4, 12, 48])
def test_invconv(ops, c):
    from normalizing_flow import InvConv2d
    m = InvConv2d(c)
    m.load_state_dict({"weight": T(ops[f"invconv{c}_w"])})
    x = cu(ops[f"invconv{c}_x"])
    ldj = z64(4)
    y, _, _ = m.transform(x, ldj, None)
    assert_close(y, ops[f"invconv{c}_y"], TOL, "y")
    assert_close(ldj, ops[f"invconv{c}_ldj"], TOL, "ldj")
    assert_close(m.invert(y), ops[f"invconv{c}_inv"], TOL, "inv")


@pytest.mark.parametrize("c", [3, 20, 96, 192])
def test_invconv_generic_channels_vs_oracle(c):
    """Channel counts without a templated kernel (3 = reference unit test, 96/192 = CelebA L5 levels)."""
    from normalizing_flow import InvConv2d
    m = InvConv2d(c)
    w = T(synth_tensor("g.invconv2d.weight", (c, c, 1, 1), 9))
    m.load_state_dict({"weight": w})
    x = T(synth_input(f"gen{c}", (2, c, 4, 4), 9))
    ldj = z64(2)
    y, _, _ = m.transform(x.to(DEV), ldj, None)
    yo, lo = O.invconv_fwd(x, w, torch.zeros(2, dtype=torch.float64))
    assert_close(y, yo, TOL, "y")
    assert_close(ldj, lo, TOL, "ldj")
    assert_close(m.invert(y), x, 1e-3, "roundtrip")


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("tag,c,nf", [("a", 4, 64), ("b", 12, 64), ("c", 4, 512), ("d", 48, 512)])
def test_coupling(ops, tag, c, nf, compute):
    from normalizing_flow import AffineCoupling
    m = AffineCoupling(c, n_features=nf, compute=compute)
    sd = {k: T(synth_tensor("flows.0.affcoupling." + k, v.shape, 2)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    x = cu(ops[f"coup{tag}_x"])
    ldj = z64(2)
    with torch.no_grad():
        y, _, _ = m.transform(x, ldj, None)
        inv = m.invert(y)
    tol = TOL if compute == "f32" else 2e-2          # bf16 MFMA inputs: 8-bit mantissa, stated separately
    assert_close(y, ops[f"coup{tag}_y"], tol, "y")
    assert_close(ldj, ops[f"coup{tag}_ldj"], tol, "ldj")
    assert_close(inv, ops[f"coup{tag}_inv"], tol, "inv")
    assert torch.equal(y[:, : c // 2], x[:, : c // 2])          # x_a passes through untouched


@pytest.mark.parametrize("c", [4, 24])
def test_split(ops, c):
    from normalizing_flow import Split
    m = Split(c)
    sd = {k: T(synth_tensor("blocks.0.split." + k, v.shape, 3)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    x = cu(ops[f"split{c}_x"])
    logp = z64(2)
    with torch.no_grad():
        y, _, z, _ = m.transform(x, None, logp)
        assert torch.equal(y.cpu(), T(ops[f"split{c}_y"])) and torch.equal(z.cpu(), T(ops[f"split{c}_z"]))
        assert_close(logp, ops[f"split{c}_logp"], TOL, "logp")
        assert torch.equal(m.invert(y, z).cpu(), T(ops[f"split{c}_inv_given"]))
        inv = m.invert(y, None, temperature=0.7, eps=cu(ops[f"split{c}_eps"]))
        assert_close(inv, ops[f"split{c}_inv_sampled"], TOL, "sampled")
        y2, _, z2, lp = m.transform(x, None, None)               # logp=None skips the prior (NFBackbone path)
        assert lp is None and torch.equal(z2, z)


@pytest.mark.parametrize("c", [16, 48])
def test_gaussian_prior(ops, c):
    from normalizing_flow import GaussianPrior
    m = GaussianPrior(c)
    sd = {k: T(synth_tensor(k, v.shape, 4)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    x = cu(ops[f"gprior{c}_x"])
    assert_close(m.compute_log_prob(x), ops[f"gprior{c}_logp"], TOL, "logp")
    s = m.sample((3, c, 4, 4), temperature=0.8, eps=cu(ops[f"gprior{c}_eps"]))
    assert_close(s, ops[f"gprior{c}_sample"], TOL, "sample")
    assert m.sample((5, c, 4, 4)).shape == (5, c, 4, 4)


def test_pre_post_bpd(ops):
    from normalizing_flow import preprocess_batch, postprocess_batch, calculate_loss
    u8 = cu(ops["pp_u8"])
    b01 = (T(ops["pp_u8"]).float() / 255.0).to(DEV)        # ToTensor runs on the host in the reference's loaders
    for nb in (5, 8):
        assert torch.equal(preprocess_batch(b01, nb, 2.0 ** nb).cpu(), T(ops[f"pre{nb}"]))
        assert torch.equal(preprocess_batch(u8, nb, 2.0 ** nb).cpu(), T(ops[f"pre{nb}"]))
    assert torch.equal(postprocess_batch(cu(ops["post_x"]), 32.0), T(ops["post5"]))
    assert_close(calculate_loss(cu(ops["bpd_ll"]), 32.0, 3072.0), ops["bpd"], 1e-12, "bpd")


def test_conv_epilogue_shapes_vs_oracle():
    """Every tile configuration (Cout 6..512), 1x1/3x3/7x7, ragged pixel counts, channel-half inputs."""
    import torch.nn.functional as F
    for (cin, cout, k, hw, b) in [(6, 512, 3, 16, 3), (512, 512, 1, 8, 2), (512, 12, 3, 16, 2), (24, 48, 3, 4, 5),
                                  (64, 64, 3, 8, 2), (2, 64, 7, 16, 1), (192, 128, 3, 8, 1), (256, 20, 1, 5, 3),
                                  (3, 30, 3, 7, 2)]:
        w = T(synth_input(f"w{cin}_{cout}_{k}", (cout, cin, k, k), 3)) / np.sqrt(cin * k * k)
        x = T(synth_input(f"x{cin}_{cout}_{k}", (b, 2 * cin, hw, hw), 3))
        add = T(synth_input("add", (cout,), 3)) * 0.1
        mul = T(synth_input("mul", (cout,), 3)) * 0.1
        ref = torch.relu((F.conv2d(x[:, :cin], w, None, padding=k // 2) + add[None, :, None, None]) *
                         torch.exp(3.0 * mul)[None, :, None, None])
        xh = K.to_nhwc(x.to(DEV))
        for comp, tol in ((K.F32, 2e-5), (K.BF16, 4e-2)):            # bf16 element-wise: observed 2.5e-2 at K = 54
            packed = K.conv_pack(w.to(DEV), comp)
            y = K.conv2d(xh[..., :cin], packed, cout, k, add=add.to(DEV), mul=mul.to(DEV), mul_exp=3.0, relu=True,
                         compute=comp)
            assert_close(K.as_nchw(y), ref, tol, f"conv {cin}->{cout} k{k} comp{comp}")
        yb = K.conv2d(xh[..., :cin], K.conv_pack(w.to(DEV), K.BF16), cout, k, compute=K.BF16, out_dtype=K.BF16)
        assert yb.dtype == torch.bfloat16
        assert_close(K.as_nchw(yb.float()), F.conv2d(x[:, :cin], w, None, padding=k // 2), 3e-2, "bf16 out")


def test_no_cpu_fallback_on_gpu_box():
    from normalizing_flow import ActNorm
    m = ActNorm(4)
    with pytest.raises(K.NfdpError):
        m.transform(torch.zeros(1, 4, 2, 2), torch.zeros(1, dtype=torch.float64), None)   # CPU tensor -> refuse


@torch.no_grad()
def test_invconv_logdets_multi_size_matches_fp64_slogdet():
    """nfdp_invconv_logdets_f32: every 1x1-conv log|det| of a forward pass in one launch (matrices of different sizes),
    against torch.slogdet in fp64 on the host and against the per-level Gauss-Jordan kernel."""
    g = torch.Generator().manual_seed(12)
    ws = []
    for c in (12, 12, 24, 48, 48, 4, 1, 96, 24):
        q, _ = torch.linalg.qr(torch.randn(c, c, generator=g))
        ws.append((q.contiguous() + 0.05 * torch.randn(c, c, generator=g)).contiguous())
    dev = [w.to("cuda") for w in ws]
    got = K.invconv_logdets(dev).cpu().double()
    ref = torch.stack([torch.slogdet(w.double())[1] for w in ws])
    assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    for w, v in zip(dev, got):
        lad, _ = K.invconv_prepare(w.unsqueeze(0).contiguous(), want_inverse=False)
        assert abs(float(lad[0]) - float(v)) <= 2e-6
    assert torch.equal(K.invconv_logdets(dev).cpu().double(), got)            # deterministic


@pytest.mark.parametrize("shape,cout,out_bf16", [((4, 16, 16, 512), 512, True), ((5, 11, 11, 512), 512, False),
                                                  ((3, 16, 16, 512), 108, False), ((2, 16, 16, 128), 192, True)])
def test_pointwise_bf16_gemm(shape, cout, out_bf16):
    """1x1 conv on bf16 activations (LDS-DMA GEMM kernel): against an fp32 matmul of the same bf16-rounded operands with the
    ActNorm-style epilogue; pixel counts that are not a multiple of the 128-pixel tile, Cout that is not a multiple of 128."""
    b, h, w, cin = shape
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(shape, generator=g).bfloat16()
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5)
    add = torch.randn(cout, generator=g) * 0.1
    mul = torch.randn(cout, generator=g) * 0.1
    ref = x.float().reshape(-1, cin) @ wt.reshape(cout, cin).bfloat16().float().t()
    ref = torch.relu((ref + add) * torch.exp(3.0 * mul)).reshape(b, h, w, cout)
    packed = K.conv_pack(wt.to(DEV), K.BF16)
    got = K.conv2d(x.to(DEV), packed, cout, 1, add=add.to(DEV), mul=mul.to(DEV), mul_exp=3.0, relu=True, compute=K.BF16,
                   out_dtype=K.BF16 if out_bf16 else K.F32)
    assert got.dtype == (torch.bfloat16 if out_bf16 else torch.float32)
    err = (got.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (6e-3 if out_bf16 else 2e-5), err
    # no epilogue, strided input view (a channel slice of a wider tensor)
    wide = torch.randn(b, h, w, cin + 64, generator=g).bfloat16().to(DEV)
    got2 = K.conv2d(wide[..., 64:], packed, cout, 1, compute=K.BF16)
    ref2 = (wide[..., 64:].float().reshape(-1, cin) @ wt.reshape(cout, cin).bfloat16().float().t().to(DEV)).reshape(b, h, w, cout)
    err2 = (got2 - ref2).abs().max().item() / ref2.abs().max().item()
    assert err2 < 2e-5, err2


@pytest.mark.parametrize("shape,cout,k", [((3, 8, 8, 64), 64, 3), ((2, 6, 10, 40), 100, 3), ((2, 4, 4, 256), 128, 1),
                                           ((1, 16, 16, 192), 128, 3)])
def test_small_fp32_conv_split_k(shape, cout, k):
    """Small fp32 convs (U-Net layers of the reverse step) take the wave-split-K kernel: against F.conv2d, with partial pixel
    tiles, Cout not a multiple of 64, zero padding at the borders, and the (acc+add)*exp(f*mul) -> ReLU epilogue."""
    import torch.nn.functional as F
    b, h, w, cin = shape
    g = torch.Generator().manual_seed(cin * k + cout)
    x = torch.randn(shape, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    add = torch.randn(cout, generator=g) * 0.1
    mul = torch.randn(cout, generator=g) * 0.1
    conv = F.conv2d(x.permute(0, 3, 1, 2), wt, padding=k // 2).permute(0, 2, 3, 1)
    packed = K.conv_pack(wt.to(DEV), K.F32)
    got = K.conv2d(x.to(DEV), packed, cout, k)
    assert_close(got.cpu(), conv, 2e-5, "plain")
    got = K.conv2d(x.to(DEV), packed, cout, k, add=add.to(DEV), mul=mul.to(DEV), mul_exp=3.0, relu=True)
    assert_close(got.cpu(), torch.relu((conv + add) * torch.exp(3.0 * mul)), 2e-5, "epilogue")
    wide = torch.randn(b, h, w, cin + 8, generator=g)
    got = K.conv2d(wide.to(DEV)[..., 8:], packed, cout, k)
    assert_close(got.cpu(), F.conv2d(wide[..., 8:].permute(0, 3, 1, 2), wt, padding=k // 2).permute(0, 2, 3, 1), 2e-5, "strided view")


@pytest.mark.parametrize("c", [65, 72, 96, 100, 128, 160, 192, 200])
def test_invconv_prepare_large_c(c):
    """log|det W| and W^-1 for 64 < C: the register-resident in-place Gauss-Jordan (C <= 192) and the global-workspace kernel
    beyond, against fp64 LAPACK; several matrices per launch, rows permuted so the implicit pivoting has work to do."""
    g = torch.Generator().manual_seed(c)
    q, _ = torch.linalg.qr(torch.randn(3, c, c, generator=g, dtype=torch.float64))
    w = (q * torch.exp(0.3 * torch.randn(3, 1, c, generator=g, dtype=torch.float64)))[:, torch.randperm(c, generator=g)]
    lad, winv = K.invconv_prepare(w.float().to(DEV), want_inverse=True)
    w32 = w.float().double()
    # numpy's LAPACK, matrix by matrix (torch's batched CPU slogdet returned garbage for this case when run late in the full suite)
    ref_lad = torch.tensor([np.linalg.slogdet(m.numpy())[1] for m in w32])
    assert float((lad.cpu().double() - ref_lad).abs().max()) < 1e-5 * max(1.0, float(ref_lad.abs().max()))
    ref_inv = torch.stack([torch.from_numpy(np.linalg.inv(m.numpy())) for m in w32])
    assert float((winv.cpu().double() - ref_inv).abs().max()) < 1e-5 * float(ref_inv.abs().max())
    lad2, none = K.invconv_prepare(w.float().to(DEV), want_inverse=False)
    assert none is None and torch.equal(lad2, lad)


def test_f32_products_on_bf16_matrix_cores():
    """nfdp_set_f32_products(1): fp32 operands split three ways into bf16 (exact), six bf16 MFMAs per product - as close to the
    fp64 result as the exact fp32 MFMA chain (both a few 1e-7), incl. padding borders, ragged tiles and the epilogue."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 9, 7, 72, generator=g) * 3.0
    wt = torch.randn(100, 72, 3, 3, generator=g) / 25.0
    add, mul = torch.randn(100, generator=g) * 0.1, torch.randn(100, generator=g) * 0.1
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), padding=1).permute(0, 2, 3, 1)
    ref_e = torch.relu((ref + add.double()) * torch.exp(3.0 * mul.double()))
    packed = K.conv_pack(wt.to(DEV), K.F32)
    errs = {}
    default_mode = K.get_f32_products()
    assert default_mode == "bf16x6"
    try:
        for mode in ("exact", "bf16x6"):
            K.set_f32_products(mode)
            got = K.conv2d(x.to(DEV), packed, 100, 3)
            got_e = K.conv2d(x.to(DEV), packed, 100, 3, add=add.to(DEV), mul=mul.to(DEV), mul_exp=3.0, relu=True)
            errs[mode] = max(float((got.cpu().double() - ref).abs().max() / ref.abs().max()),
                             float((got_e.cpu().double() - ref_e).abs().max() / ref_e.abs().max()))
    finally:
        K.set_f32_products(default_mode)
    assert errs["exact"] < 2e-6 and errs["bf16x6"] < 2e-6, errs
    # the tiled conv kernel (short K: not the split-K path) takes the same switch
    x2 = torch.randn(4, 16, 16, 12, generator=g)
    w2 = torch.randn(96, 12, 3, 3, generator=g) / 10.0
    ref2 = F.conv2d(x2.permute(0, 3, 1, 2).double(), w2.double(), padding=1).permute(0, 2, 3, 1)
    p2 = K.conv_pack(w2.to(DEV), K.F32)
    try:
        for mode in ("exact", "bf16x6"):
            K.set_f32_products(mode)
            got2 = K.conv2d(x2.to(DEV), p2, 96, 3)
            assert float((got2.cpu().double() - ref2).abs().max() / ref2.abs().max()) < 2e-6, mode
    finally:
        K.set_f32_products(default_mode)


def test_f32_split_products_keep_non_finite_operands_non_finite():
    """compute = NFDP_F32X6 (three-way bf16 splits): an inf / NaN / above-bf16-range operand goes through the high piece alone (ADVICE
    r4: v - bf16(v) = inf - inf used to make EVERY such operand a NaN source, also where the exact chain stays finite).  What the split
    form can promise: the outputs an inf / NaN operand reaches are non-finite exactly where the exact fp32 chain's are, NaN stays NaN,
    and every other output equals the clean result.  It cannot promise the SIGN of an inf: inf times the other operand's (signed)
    residual pieces gives -inf terms, so an inf output of the exact chain may be NaN here (stated in include/nfdp.h)."""
    g = torch.Generator().manual_seed(5)
    for shape_x, shape_w in (((2, 8, 8, 72), (64, 72, 3, 3)), ((2, 16, 16, 12), (96, 12, 3, 3))):      # split-K kernel, tiled kernel
        x = torch.randn(*shape_x, generator=g)
        wt = torch.randn(*shape_w, generator=g) / 20.0
        packed = K.conv_pack(wt.to(DEV), K.F32)
        default_mode = K.get_f32_products()
        try:
            for name, val in {"inf": float("inf"), "-inf": float("-inf"), "nan": float("nan")}.items():
                xv = x.clone()
                xv[1, 3, 4, 5] = val
                out = {}
                for mode in ("exact", "bf16x6"):
                    K.set_f32_products(mode)
                    out[mode] = K.conv2d(xv.to(DEV), packed, shape_w[0], 3).cpu()
                e, s = out["exact"], out["bf16x6"]
                assert torch.equal(torch.isfinite(e), torch.isfinite(s)), name
                assert torch.equal(torch.isnan(e) & torch.isnan(s), torch.isnan(e)), name          # NaN stays NaN
                fin = torch.isfinite(e)
                assert fin[0].all() and not fin[1].all() and int((~fin).sum()) <= 9 * shape_w[0], name
                assert float((e[fin] - s[fin]).abs().max()) < 1e-4 * float(e[fin].abs().max()), name
        finally:
            K.set_f32_products(default_mode)


def test_coupling_chunked_matches_per_sample_kernel():
    """Large image, small batch: the coupling update with several workgroups per sample (chunk sums of the log-det combined in
    chunk order) against the one-workgroup-per-sample kernel on the same data, forward and inverse."""
    g = torch.Generator().manual_seed(3)
    b, hw, c = 4, 64, 12
    x = torch.randn(b, hw, hw, c, generator=g).to(DEV)
    h = (torch.randn(b, hw, hw, c, generator=g) * 0.5).to(DEV)
    cst = torch.tensor([0.25, -1.5], device=DEV)
    assert K.lib().nfdp_coupling_chunked_ws_bytes(b, hw * hw, c) > 0
    y, ldj = x.clone(), torch.full((b,), 3.0, dtype=torch.float64, device=DEV)
    K.coupling_fwd(y, h, ldj, cst)
    # reference: the same kernel family on a geometry that stays per-sample (split the image rows into 4 'samples' of 16 rows)
    y2 = x.clone().reshape(b * 4, 16, hw, c)
    ldj2 = torch.zeros(b * 4, dtype=torch.float64, device=DEV)
    assert K.lib().nfdp_coupling_chunked_ws_bytes(b * 4, 16 * hw, c) == 0
    K.coupling_fwd(y2, h.reshape(b * 4, 16, hw, c), ldj2, None)
    assert torch.equal(y, y2.reshape(b, hw, hw, c))
    ref = 3.0 + (0.25 - 1.5) + ldj2.reshape(b, 4).sum(dim=1)
    assert float((ldj - ref).abs().max()) < 1e-9 * float(ref.abs().max())
    xr = y.clone()
    K.coupling_inv(xr, h)
    assert float((xr - x).abs().max()) < 1e-4


@torch.no_grad()
def test_inverses_of_all_levels_in_one_launch_equal_the_per_level_launches():
    """nfdp_invconv_inverses_f32 (W^-1 of matrices of different sizes, one wave each, ONE launch: the 1x1 convs of all levels ahead
    of Glow.invert) against nfdp_invconv_prepare_f32 level by level: the same Gauss-Jordan -> bit-identical; and against fp64."""
    g = torch.Generator().manual_seed(5)
    groups = [[(torch.linalg.qr(torch.randn(c, c, generator=g))[0] + 0.1 * torch.randn(c, c, generator=g)).contiguous().to(DEV) for _ in range(k)]
              for c, k in ((12, 16), (24, 16), (48, 16), (4, 3), (8, 1), (16, 2))]
    assert all(K.invconv_inverses_supported(gr[0].shape[-1]) for gr in groups) and not K.invconv_inverses_supported(96)
    outs = K.invconv_inverses(groups)
    for gr, out in zip(groups, outs):
        _, ref = K.invconv_prepare(torch.stack(gr), want_inverse=True)
        assert torch.equal(out, ref)
        truth = torch.linalg.inv(torch.stack(gr).double().cpu())
        assert relerr(out, truth) < 1e-6
