"""Backward kernels and the differentiable training path against torch autograd: per op on the CPU
(plain torch fp32 reference of the same op) and, for the whole Glow + Gaussian prior bits/dim, against autograd through
the oracle on the same weights.  fp32, tolerance 1e-4 relative to the largest entry of each gradient tensor."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import nfdp_hip as K
from oracle import nf_oracle as O
from tests.util import T, assert_close, relerr
from tests.golden.synth import synth_tensor, synth_input, DEEP

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("b,hw,cin,cout,k", [(2, 8, 6, 512, 3), (3, 4, 512, 512, 1), (2, 8, 512, 12, 3), (5, 4, 24, 48, 3),
                                             (1, 16, 3, 7, 3), (4, 4, 48, 48, 1), (2, 16, 64, 96, 3), (7, 2, 130, 33, 1)])
def test_conv_wgrad_and_dgrad(b, hw, cin, cout, k):
    x = rnd(b, cin, hw, hw, seed=1).requires_grad_()
    w = rnd(cout, cin, k, k, seed=2, scale=0.1).requires_grad_()
    dy = rnd(b, cout, hw, hw, seed=3)
    F.conv2d(x, w, padding=k // 2).backward(dy)
    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dyh = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw = K.conv_wgrad(xh, dyh, k)
    assert_close(dw, w.grad, TOL, "wgrad")
    wt = K.conv_flip_weights(w.detach().to(DEV).contiguous())
    dx = K.conv2d(dyh, K.conv_pack(wt, K.F32), cin, k, compute=K.F32, out_dtype=K.F32)
    assert_close(dx.permute(0, 3, 1, 2), x.grad, TOL, "dgrad")
    # pixel-strided operands (channel slices of wider tensors), as the coupling network sees them
    wide = torch.zeros(b, hw, hw, cin + 5, device=DEV)
    wide[..., 2:2 + cin] = xh
    assert torch.equal(K.conv_wgrad(wide[..., 2:2 + cin], dyh, k), dw)
    # determinism
    assert torch.equal(K.conv_wgrad(xh, dyh, k), dw)


@pytest.mark.parametrize("c,f,relu", [(12, 1.0, False), (512, 1.0, True), (48, 3.0, False), (130, 3.0, True)])
def test_scale_bwd(c, f, relu):
    u = rnd(3, 5, 7, c, seed=4).requires_grad_()
    add = rnd(c, seed=5, scale=0.3).requires_grad_()
    mul = rnd(c, seed=6, scale=0.2).requires_grad_()
    y = (u + add) * torch.exp(f * mul)
    if relu:
        y = torch.relu(y)
    dy = rnd(3, 5, 7, c, seed=7)
    y.backward(dy)
    dc, dmul, dadd = K.scale_bwd(dy.to(DEV), y.detach().to(DEV), mul.detach().to(DEV), f, relu)
    assert_close(dc, u.grad, TOL, "dc")
    assert_close(dmul, mul.grad, TOL, "dmul")
    assert_close(dadd, add.grad, TOL, "dadd")
    _, none, colsum = K.scale_bwd(dy.to(DEV), want_dc=False)
    assert none is None
    assert_close(colsum, dy.reshape(-1, c).sum(0), TOL, "column sum")


@pytest.mark.parametrize("c", [4, 12, 48])
def test_coupling_and_gauss_bwd(c):
    b, hw = 3, 4
    x = rnd(b, c, hw, hw, seed=8).requires_grad_()
    h = rnd(b, c, hw, hw, seed=9, scale=0.7).requires_grad_()
    gl = torch.tensor([0.3, -1.1, 0.8], dtype=torch.float64)
    xa, xb = x.chunk(2, 1)
    log_s, t = h.chunk(2, 1)
    s = torch.sigmoid(log_s + 2.0)
    y = torch.cat([xa, (xb + t) * s], 1)
    inc = torch.log(s + 1e-6).flatten(1).sum(1)
    dy = rnd(b, c, hw, hw, seed=10)
    ((y * dy).sum() + (inc.double() * gl).sum()).backward()
    nh = lambda a: a.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dx, dh = K.coupling_bwd(nh(x), nh(h), nh(dy), gl.to(DEV))
    assert_close(dx.permute(0, 3, 1, 2), x.grad, TOL, "coupling dx")
    assert_close(dh.permute(0, 3, 1, 2), h.grad, TOL, "coupling dh")
    # Gaussian log-density, both h layouts
    z = rnd(b, c, hw, hw, seed=11).requires_grad_()
    for mode in (0, 1):
        hh = (rnd(b, 2 * c, hw, hw, seed=12, scale=0.4) if mode == 0 else rnd(2 * c, seed=13, scale=0.4)).requires_grad_()
        z.grad = None
        mean, logs = (hh.chunk(2, 1) if mode == 0 else [v.view(1, c, 1, 1) for v in hh.chunk(2, 0)])
        lp = (-0.5 * (np.log(2 * np.pi) + 2.0 * logs + (z - mean) ** 2 * torch.exp(-2.0 * logs))).flatten(1).sum(1)
        (lp.double() * gl).sum().backward()
        hdev = nh(hh) if mode == 0 else hh.detach().to(DEV)
        dz, dhh = K.gauss_logp_bwd(nh(z), hdev, mode, gl.to(DEV))
        assert_close(dz.permute(0, 3, 1, 2), z.grad, TOL, f"gauss dz mode {mode}")
        if mode == 0:
            assert_close(dhh.permute(0, 3, 1, 2), hh.grad, TOL, "gauss dh")
        else:
            assert_close(K.scale_bwd(dhh, want_dc=False)[2], hh.grad, TOL, "gauss dh (per channel)")


def build(C, L, Kf, seed):
    from normalizing_flow import Glow, GaussianPrior
    flow, prior = Glow(in_channel=C, L=L, K=Kf, learn_prior_mean_logs=True, compute="f32"), GaussianPrior(C * 2 ** (L + 1), True)
    fsd = {k: T(synth_tensor(k, v.shape, seed)) for k, v in flow.state_dict().items()}
    psd = {k: T(synth_tensor(k, v.shape, seed + 1)) for k, v in prior.state_dict().items()}
    flow.load_state_dict(fsd, strict=True)
    prior.load_state_dict(psd, strict=True)
    return flow.to(DEV), prior.to(DEV), fsd, psd


@pytest.mark.parametrize("C,L,Kf,B,size", [(3, 3, 2, 2, 32), (1, 2, 2, 3, 16)])
def test_glow_training_gradients_match_oracle_autograd(C, L, Kf, B, size):
    """loss = bits/dim of Glow + Gaussian prior (trainer.py:152-164): every parameter gradient of the HIP training path
    against autograd through the CPU oracle on the same weights and inputs."""
    from normalizing_flow import calculate_loss, initialize_with_zeros
    flow, prior, fsd, psd = build(C, L, Kf, 31)
    x = T(synth_input("train_x", (B, C, size, size), 5)) * 0.4
    # oracle
    fo = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fsd.items()}
    po = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in psd.items()}
    _, _, _, bpd_ref = O.fwd_nll(x, fo, po, L, Kf, img_size=size)
    bpd_ref.backward()
    # HIP training path, written like the reference's loop body
    flow.train(); prior.train()
    ll, logp = initialize_with_zeros(2, B, torch.device(DEV))
    latents, ll, logp = flow.transform(x.to(DEV), ll, logp)
    logp = logp + prior.compute_log_prob(latents[-1])
    loss = calculate_loss(ll + logp, 2.0 ** 5, size * size * 3.0)
    assert loss.requires_grad
    assert_close(loss, bpd_ref.detach(), 1e-5, "bits/dim of the training path")
    loss.backward()
    worst = ("", 0.0)
    for mod, ref in ((flow, fo), (prior, po)):
        for k, p in mod.named_parameters():
            g_ref = ref[k].grad
            assert p.grad is not None, f"no gradient for {k}"
            assert torch.isfinite(p.grad).all(), k
            if g_ref is None or float(g_ref.abs().max()) == 0.0:
                assert float(p.grad.abs().max()) <= 1e-12, f"{k}: expected a zero gradient"
                continue
            e = relerr(p.grad, g_ref)
            if e > worst[1]:
                worst = (k, e)
            assert e <= TOL, f"grad {k}: rel err {e:.3e}"
    print("worst gradient", worst)
    # inference path is unchanged by the training path having run
    with torch.no_grad():
        ll2, lp2 = initialize_with_zeros(2, B, torch.device(DEV))
        _, ll2, lp2 = flow.transform(x.to(DEV), ll2, lp2)
        assert_close(ll2, ll.detach(), 1e-6, "ldj train vs inference path")
        assert_close(lp2, (logp - prior.compute_log_prob(latents[-1])).detach(), 1e-5, "logp train vs inference path")


def test_reference_style_training_step_runs():
    """zero_grad -> backward -> clip_grad_value_ -> clip_grad_norm_ -> Adam step (trainer.py:163-167) lowers the loss."""
    from normalizing_flow import calculate_loss, initialize_with_zeros, init_optimizer
    flow, prior, _, _ = build(3, 2, 2, 41)
    opt = init_optimizer("adam", [{"params": list(flow.parameters())}, {"params": list(prior.parameters())}], lr=1e-3)
    x = (T(synth_input("train_x2", (8, 3, 16, 16), 6)) * 0.4).to(DEV)
    losses = []
    for _ in range(4):
        ll, logp = initialize_with_zeros(2, 8, torch.device(DEV))
        latents, ll, logp = flow.transform(x, ll, logp)
        logp = logp + prior.compute_log_prob(latents[-1])
        loss = calculate_loss(ll + logp, 32.0, 16 * 16 * 3.0)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(flow.parameters(), 1)
        torch.nn.utils.clip_grad_norm_(flow.parameters(), 1)
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("adamw,wd", [(False, 0.0), (False, 0.01), (True, 0.01)])
def test_fused_adam_matches_torch(adamw, wd):
    """nfdp_fused_adam_step_f32 (value clip + global-norm clip + Adam/AdamW on flat buffers) against
    clip_grad_value_ + clip_grad_norm_ + torch.optim.Adam/AdamW, five steps, clipped and unclipped parameter groups."""
    from nfdp_hip.optim import FusedAdam
    shapes_a, shapes_b = [(512, 6, 3, 3), (512,), (12, 12, 1, 1), (33, 7)], [(96,), (5, 5)]
    g = torch.Generator().manual_seed(3)
    mk = lambda shapes: [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.5).to(DEV)) for s in shapes]
    pa, pb = mk(shapes_a), mk(shapes_b)
    ra, rb = [torch.nn.Parameter(p.detach().clone()) for p in pa], [torch.nn.Parameter(p.detach().clone()) for p in pb]
    fused = FusedAdam(clipped=pa, unclipped=pb, lr=1e-2, weight_decay=wd, adamw=adamw, clip_value=1.0, max_norm=1.0)
    Ref = torch.optim.AdamW if adamw else torch.optim.Adam
    ref = Ref([{"params": ra}, {"params": rb}], lr=1e-2, weight_decay=wd)
    for step in range(5):
        grads = [torch.randn(p.shape, generator=g).to(DEV) * (3.0 if step % 2 else 0.3) for p in pa + pb]
        fused.zero_grad()
        ref.zero_grad()
        for p, r, gr in zip(pa + pb, ra + rb, grads):
            (p * gr).sum().backward()                               # autograd accumulates into the flat-buffer views
            r.grad = gr.clone()
        torch.nn.utils.clip_grad_value_(ra, 1)
        tn = torch.nn.utils.clip_grad_norm_(ra, 1)
        ref.step()
        fused.step()
        assert_close(fused.grad_norm, tn.reshape(1), 1e-5, "global norm")
        for p, r in zip(pa + pb, ra + rb):
            assert_close(p, r, 2e-5, f"step {step} parameter {tuple(p.shape)}")
    assert all(p.data_ptr() >= fused.flat_param.data_ptr() for p in pa + pb)


def test_training_step_with_fused_optimizer_updates_kernel_weight_caches():
    """The fused optimizer updates parameters through its flat buffer; the packed-weight caches of the inference path
    must follow (parameter epoch), i.e. the no-grad forward after a step equals the oracle on the updated weights."""
    from normalizing_flow import calculate_loss, initialize_with_zeros
    from nfdp_hip.optim import FusedAdam
    flow, prior, _, _ = build(3, 2, 2, 51)
    opt = FusedAdam(clipped=flow.parameters(), unclipped=prior.parameters(), lr=1e-3, clip_value=1, max_norm=1)
    x = (T(synth_input("train_x3", (4, 3, 16, 16), 7)) * 0.4).to(DEV)
    dev = torch.device(DEV)

    def nll_nograd():
        with torch.no_grad():
            ll, lp = initialize_with_zeros(2, 4, dev)
            lat, ll, lp = flow.transform(x, ll, lp)
            return float(calculate_loss(ll + lp + prior.compute_log_prob(lat[-1]), 32.0, 16 * 16 * 3.0))

    before = nll_nograd()
    losses = []
    for _ in range(3):
        ll, lp = initialize_with_zeros(2, 4, dev)
        lat, ll, lp = flow.transform(x, ll, lp)
        loss = calculate_loss(ll + lp + prior.compute_log_prob(lat[-1]), 32.0, 16 * 16 * 3.0)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    after = nll_nograd()
    assert abs(losses[0] - before) <= 1e-5 * abs(before)
    assert all(np.isfinite(losses)) and abs(after - before) > 1e-4, (before, losses, after)   # the weights did move
    fsd = {k: v.detach().cpu() for k, v in flow.state_dict().items()}
    psd = {k: v.detach().cpu() for k, v in prior.state_dict().items()}
    _, _, _, bpd = O.fwd_nll(x.cpu(), fsd, psd, 2, 2, img_size=16)
    assert abs(float(bpd) - after) <= 2e-5 * abs(after), (float(bpd), after)


def test_training_step_matches_reference_goldens():
    """The reference's own training step on the glow_N weights/inputs (tests/golden/train_N.npz: generated by running
    the reference): gradients before the clips, gradients after clip_grad_value_(1) + clip_grad_norm_(1), the clipped
    norm, and the parameters after one Adam step - HIP training path + nfdp_fused_adam_step_f32."""
    import json
    from normalizing_flow import Glow, GaussianPrior, calculate_loss, initialize_with_zeros
    from nfdp_hip.optim import FusedAdam
    from tests.util import load_golden, meta_of, synth_sd
    from tests.test_oracle_golden import close_summary
    g = load_golden("train_N.npz")
    m = meta_of(g)
    flow = Glow(in_channel=m["C"], L=m["L"], K=m["K"], compute="f32")
    prior = GaussianPrior(2 ** (m["L"] + 1) * m["C"])
    flow.load_state_dict(synth_sd(m["flow_spec"], m["seed"]))
    prior.load_state_dict(synth_sd(m["prior_spec"], m["seed"]))
    flow.to(DEV); prior.to(DEV)
    opt = FusedAdam(clipped=flow.parameters(), unclipped=prior.parameters(), lr=1e-3, clip_value=1, max_norm=1)
    x = T(load_golden("glow_N.npz")["x"]).to(DEV)
    ll, logp = initialize_with_zeros(2, m["B"], torch.device(DEV))
    lat, ll, logp = flow.transform(x, ll, logp)
    logp = logp + prior.compute_log_prob(lat[-1])
    loss = calculate_loss(ll + logp, 32.0, 32 * 32 * 3.0)
    assert_close(loss, g["bpd"], 1e-5, "bpd")
    opt.zero_grad()
    loss.backward()
    names, pnames = dict(flow.named_parameters()), dict(prior.named_parameters())
    keys, pkeys = json.loads(str(g["keys"])), json.loads(str(g["pkeys"]))
    for i, k in enumerate(keys):
        close_summary(names[k].grad, g[f"g{i}"], TOL, f"grad {k}")
    for i, k in enumerate(pkeys):
        if float(g[f"pg{i}"][-1]) > 0:
            close_summary(pnames[k].grad, g[f"pg{i}"], TOL, f"prior grad {k}")
    sumsq = sum(float((p.grad.double() ** 2).sum()) for p in flow.parameters())
    assert abs(sumsq - float(g["sumsq_all"])) <= 2e-4 * float(g["sumsq_all"])
    opt.step()                                        # value clip, norm clip (gradients scaled in place), Adam
    assert_close(opt.grad_norm, g["total_norm_after_value_clip"].reshape(1), TOL, "clipped global norm")
    for i, k in enumerate(keys):
        close_summary(names[k].grad, g[f"c{i}"], TOL, f"clipped grad {k}")
        close_summary(names[k], g[f"p{i}"], TOL, f"updated parameter {k}")
    for i, k in enumerate(pkeys):
        close_summary(pnames[k], g[f"pp{i}"], TOL, f"updated prior parameter {k}")


@pytest.mark.parametrize("b,hw,cin,cout,k", [(2, 8, 512, 512, 1), (2, 8, 12, 512, 3), (3, 4, 512, 48, 3), (5, 4, 24, 512, 3),
                                             (2, 16, 64, 96, 3), (7, 2, 132, 36, 1)])
def test_conv_wgrad_bf16(b, hw, cin, cout, k):
    """bf16-MFMA weight gradient against the fp32 autograd result: operands rounded to bf16, fp32 accumulation."""
    x = rnd(b, cin, hw, hw, seed=21).requires_grad_()
    w = rnd(cout, cin, k, k, seed=22, scale=0.1).requires_grad_()
    dy = rnd(b, cout, hw, hw, seed=23)
    F.conv2d(x, w, padding=k // 2).backward(dy)
    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dyh = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    K.start_trace()
    dw = K.conv_wgrad(xh, dyh, k, compute=K.BF16)
    labels = [e["label"] for e in K.stop_trace()]
    assert any(lb.startswith("conv_wgrad_bf16") for lb in labels), labels
    assert_close(dw, w.grad, 2e-2, "bf16 wgrad")            # bf16 operands, element-wise: observed 1.0-1.3e-2
    # exact when the operands are representable in bf16
    xq, dq = xh.bfloat16().float(), dyh.bfloat16().float()
    assert_close(K.conv_wgrad(xq, dq, k, compute=K.BF16), K.conv_wgrad(xq, dq, k), 1e-5, "bf16 kernel on bf16-exact data")
    assert torch.equal(K.conv_wgrad(xh, dyh, k, compute=K.BF16), dw)                 # deterministic
    # pixel-strided operand (channel slice) with aligned stride
    if cin % 8 == 0:
        wide = torch.zeros(b, hw, hw, 2 * cin, device=DEV)
        wide[..., cin:] = xh
        assert torch.equal(K.conv_wgrad(wide[..., cin:], dyh, k, compute=K.BF16), dw)


def test_glow_training_in_bf16_mode_tracks_the_fp32_gradients():
    """compute="bf16": the coupling networks' forward, data-gradient and weight-gradient GEMMs run on bf16 MFMA (fp32
    activations and accumulation).  Gradients stay within bf16 round-off of autograd through the fp32 oracle - measured in
    the Frobenius norm per tensor: single entries move more when a hidden unit's ReLU mask flips - and the reference-style
    loop still lowers the loss."""
    from normalizing_flow import calculate_loss, initialize_with_zeros
    from nfdp_hip.optim import FusedAdam
    from tests.golden.synth import DEEP
    B = 8
    flow, prior, _, _ = build(3, 3, 2, 31)
    fsd = {k: T(synth_tensor(k, v.shape, 31, DEEP)) for k, v in flow.state_dict().items()}     # trained-like weights
    psd = {k: T(synth_tensor(k, v.shape, 32, DEEP)) for k, v in prior.state_dict().items()}
    flow.load_state_dict(fsd); prior.load_state_dict(psd)
    flow.set_compute("bf16")
    x = T(synth_input("train_x", (B, 3, 32, 32), 5)) * 0.4
    fo = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fsd.items()}
    po = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in psd.items()}
    _, _, _, bpd_ref = O.fwd_nll(x, fo, po, 3, 2, img_size=32)
    bpd_ref.backward()
    ll, logp = initialize_with_zeros(2, B, torch.device(DEV))
    K.start_trace()
    latents, ll, logp = flow.transform(x.to(DEV), ll, logp)
    loss = calculate_loss(ll + logp + prior.compute_log_prob(latents[-1]), 32.0, 32 * 32 * 3.0)
    loss.backward()
    labels = [e["label"] for e in K.stop_trace()]
    assert any(lb.startswith("conv_wgrad_bf16") for lb in labels)
    assert any("conv_kernel<bf16" in lb or lb.startswith("level_fused_train") for lb in labels)
    assert_close(loss, bpd_ref.detach(), 2e-3, "bits/dim (bf16 coupling networks)")

    def fro(a, b):
        a, b = a.detach().double().cpu(), b.double()
        return float((a - b).norm() / b.norm())
    errs = {k: fro(p.grad, fo[k].grad) for k, p in flow.named_parameters() if float(fo[k].grad.abs().max()) > 0}
    worst = max(errs.items(), key=lambda kv: kv[1])
    print("bf16 training gradients: median rel Frobenius err", float(np.median(list(errs.values()))), "worst", worst)
    assert worst[1] <= 8e-2, worst
    assert float(np.median(list(errs.values()))) <= 1e-2
    opt = FusedAdam(clipped=flow.parameters(), unclipped=prior.parameters(), lr=2e-4, clip_value=1, max_norm=1)
    losses = []
    for _ in range(4):
        ll, logp = initialize_with_zeros(2, B, torch.device(DEV))
        latents, ll, logp = flow.transform(x.to(DEV), ll, logp)
        loss = calculate_loss(ll + logp + prior.compute_log_prob(latents[-1]), 32.0, 32 * 32 * 3.0)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("b,hw,cin,cout,relu", [(2, 8, 512, 12, False), (3, 4, 512, 48, True), (2, 16, 512, 6, False), (1, 5, 128, 3, True)])
def test_conv3x3_narrow_matches_conv(b, hw, cin, cout, relu):
    """Narrow 3x3 conv = pointwise GEMM to tap contributions + tap shift-sum, against F.conv2d with the same epilogue."""
    x = rnd(b, cin, hw, hw, seed=41)
    w = rnd(cout, cin, 3, 3, seed=42, scale=0.05)
    add, mul = rnd(cout, seed=43, scale=0.2), rnd(cout, seed=44, scale=0.1)
    ref = (F.conv2d(x, w, padding=1) + add.view(1, -1, 1, 1)) * torch.exp(3.0 * mul).view(1, -1, 1, 1)
    if relu:
        ref = torch.relu(ref)
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    got = K.conv3x3_narrow(xh, w.to(DEV), add=add.to(DEV), mul=mul.to(DEV), mul_exp=3.0, relu=relu, compute=K.F32)
    assert_close(got.permute(0, 3, 1, 2), ref, 1e-5, "narrow conv f32")
    gb = K.conv3x3_narrow(xh, w.to(DEV), add=add.to(DEV), mul=mul.to(DEV), mul_exp=3.0, relu=relu, compute=K.BF16)
    assert_close(gb.permute(0, 3, 1, 2), ref, 2e-2, "narrow conv bf16")
    plain = K.conv3x3_narrow(xh, w.to(DEV), compute=K.F32)
    assert_close(plain.permute(0, 3, 1, 2), F.conv2d(x, w, padding=1), 1e-5, "no epilogue")


def test_conv_wgrad_bf16_on_channel_slice_not_multiple_of_4():
    """x_a = the first 6 of 12 channels (level 0 of a 3-channel flow): the bf16 kernel reads 8 channels of the same pixel rows
    and the 2 extra input columns are dropped."""
    x = rnd(2, 12, 8, 8, seed=51).requires_grad_()
    w = rnd(512, 6, 3, 3, seed=52, scale=0.1).requires_grad_()
    dy = rnd(2, 512, 8, 8, seed=53)
    F.conv2d(x[:, :6], w, padding=1).backward(dy)
    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    K.start_trace()
    dw = K.conv_wgrad(xh[..., :6], dy.permute(0, 2, 3, 1).contiguous().to(DEV), 3, compute=K.BF16)
    assert any(e["label"].startswith("conv_wgrad_bf16") for e in K.stop_trace())
    assert tuple(dw.shape) == (512, 6, 3, 3)
    assert_close(dw, w.grad, 2e-2, "bf16 wgrad on a 6-of-12 channel slice")


def test_mixed_dtype_scale_bwd_and_wgrad():
    """bf16 storage of the hidden activations / gradients: the kernels must give what the fp32 kernels give on the same
    (bf16-representable) values, up to the rounding of the bf16 dc output."""
    c = 512
    dy = rnd(2, 4, 4, c, seed=61).bfloat16()
    y = torch.relu(rnd(2, 4, 4, c, seed=62)).bfloat16()
    mul = rnd(c, seed=63, scale=0.2)
    ref = K.scale_bwd(dy.float().to(DEV), y.float().to(DEV), mul.to(DEV), 1.0, True)
    got = K.scale_bwd(dy.to(DEV), y.to(DEV), mul.to(DEV), 1.0, True)
    assert got[0].dtype == torch.bfloat16
    assert_close(got[0].float(), ref[0], 4e-3, "dc (bf16 storage)")
    assert_close(got[1], ref[1], 1e-5, "dmul")
    assert_close(got[2], ref[2], 1e-5, "dadd")
    mixed = K.scale_bwd(dy.float().to(DEV), y.to(DEV), mul.to(DEV), 1.0, True)
    assert mixed[0].dtype == torch.bfloat16
    assert_close(mixed[0].float(), ref[0], 4e-3, "dc (fp32 dy, bf16 y)")
    x = rnd(2, 8, 8, 512, seed=64).bfloat16().to(DEV)
    d = rnd(2, 8, 8, 512, seed=65).bfloat16().to(DEV)
    exact = K.conv_wgrad(x.float(), d.float(), 1)
    for xx, dd in ((x, d), (x.float(), d), (x, d.float())):
        assert_close(K.conv_wgrad(xx, dd, 1, compute=K.BF16), exact, 1e-5, f"wgrad {xx.dtype}/{dd.dtype}")


def test_im2col_form_of_the_narrow_input_conv():
    """The C/2 -> 512 conv of the bf16 training path as explicit im2col + pointwise GEMM: the im2col matrix against unfold, and
    forward / weight gradient against the implicit-GEMM kernels on the same bf16-rounded operands."""
    import torch.nn.functional as F
    for c, hw in ((6, 16), (12, 8), (24, 4)):
        x = rnd(16, hw, hw, c, seed=70 + c).to(DEV)
        col = K.im2col3x3(x)
        assert col.shape[-1] == K.im2col_kpad(c) and col.dtype == torch.bfloat16
        ref = F.unfold(x.permute(0, 3, 1, 2), 3, padding=1).reshape(16, c, 9, hw, hw).permute(0, 3, 4, 2, 1).reshape(16, hw, hw, 9 * c)
        assert torch.equal(col[..., :9 * c], ref.bfloat16()) and not col[..., 9 * c:].any()
        w = rnd(512, c, 3, 3, seed=71 + c, scale=0.1).to(DEV)
        y_ref = K.conv2d(x, K.conv_pack(w, K.BF16), 512, 3, compute=K.BF16)
        y = K.conv2d(col, K.conv_pack(w, K.BF16), 512, 1, compute=K.BF16)      # the packed 3x3 weight is the 1x1 weight for these rows
        assert_close(y, y_ref, 1e-5, f"forward C={c}")
        dc = rnd(16, hw, hw, 512, seed=72 + c).bfloat16().to(DEV)
        dw = K.conv_wgrad(col, dc, 1, compute=K.BF16)[:, :9 * c, 0, 0].reshape(512, 3, 3, c).permute(0, 3, 1, 2)
        dw_ref = K.conv_wgrad(x.bfloat16().float(), dc.float(), 3)
        assert_close(dw, dw_ref, 1e-5, f"weight gradient C={c}")


def test_fused_adam_replayed_in_a_graph_takes_real_adam_steps():
    """The optimizer's step count and learning rate live on the device, so a step captured in a hipGraph replays with the
    right bias corrections (host-side scalars would freeze at capture time): 2 eager + 4 replayed steps == 6 eager steps,
    against torch.optim.Adam as well; a learning-rate change between replays is picked up."""
    from nfdp_hip.optim import FusedAdam
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 7), (33,)]
    mk = lambda: [torch.nn.Parameter((torch.randn(*s, generator=torch.Generator().manual_seed(1 + i)) * 0.5).to(DEV)) for i, s in enumerate(shapes)]
    pa, pb, pr = mk(), mk(), mk()
    grads = [[(torch.randn(*s, generator=g) * 0.3).to(DEV) for s in shapes] for _ in range(6)]
    a, b = FusedAdam(pa, lr=1e-2, max_norm=1.0), FusedAdam(pb, lr=1e-2, max_norm=1.0)
    ref = torch.optim.Adam(pr, lr=1e-2)

    def set_grads(ps, gs):
        for p, gr in zip(ps, gs):
            p.grad.copy_(gr)

    for t in range(6):                                             # all eager
        if t == 4:
            ref.param_groups[0]["lr"] = 5e-3
            a.lr = 5e-3
        set_grads(pa, grads[t]); a.step()
        for p, gr in zip(pr, grads[t]):
            p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(pr, 1.0)
        ref.step()
    static = [torch.zeros_like(x) for x in grads[0]]
    for t in range(2):
        set_grads(pb, grads[t]); b.step()
    graph = torch.cuda.CUDAGraph()
    for s_, gr in zip(static, grads[2]):
        s_.copy_(gr)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):                                  # captured step t = 2 (runs once at capture? no: capture only records)
        set_grads(pb, static); b.step()
    for t in range(2, 6):
        for s_, gr in zip(static, grads[t]):
            s_.copy_(gr)
        if t == 4:
            b.lr = 5e-3
            b._hyper[0:1].copy_(torch.tensor([5e-3], dtype=torch.float64))     # what step() does on an lr change, outside the graph
            b._lr_uploaded = 5e-3
        graph.replay()
    assert a.step_count == 6 and b.step_count == 6
    for x, y, r in zip(pa, pb, pr):
        assert_close(y, x, 1e-6, "replayed vs eager FusedAdam")
        assert_close(x, r, 3e-5, "FusedAdam vs torch Adam")


def test_graphed_training_step_then_eager_eval_sees_the_new_weights():
    """A training step replayed from a hipGraph changes the parameters behind torch's version counters; Graphed(mutates_params=True)
    bumps the parameter epoch so the eager forward afterwards repacks its weights: its bits/dim equals a fresh model's with the
    same parameter values."""
    from nfdp_hip.graphs import Graphed
    from nfdp_hip.optim import FusedAdam
    from normalizing_flow import calculate_loss, initialize_with_zeros
    from tests.golden.synth import DEEP
    flow, prior, _, _ = build(3, 3, 1, 41)
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 41, DEEP)) for k, v in flow.state_dict().items()})
    prior.load_state_dict({k: T(synth_tensor(k, v.shape, 42, DEEP)) for k, v in prior.state_dict().items()})
    flow.set_compute("bf16")
    flow.to(DEV); prior.to(DEV)
    opt = FusedAdam(clipped=flow.parameters(), unclipped=prior.parameters(), lr=1e-3, clip_value=1, max_norm=1)
    x = (T(synth_input("gx", (8, 3, 32, 32), 6)) * 0.4).to(DEV)

    def bpd(xb):
        ll, logp = initialize_with_zeros(2, xb.shape[0], torch.device(DEV))
        latents, ll, logp = flow.transform(xb, ll, logp)
        return calculate_loss(ll + logp + prior.compute_log_prob(latents[-1]), 32.0, 32 * 32 * 3.0)

    def train_step(xb):
        loss = bpd(xb)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.detach()

    with torch.no_grad():
        before = float(bpd(x))
    graphed = Graphed(train_step, [x], warmup=1, mutates_params=True)
    for _ in range(3):
        graphed(x)
    with torch.no_grad():
        after = float(bpd(x))                                      # eager, through the weight caches
        sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
        psd = {k: v.detach().clone() for k, v in prior.state_dict().items()}
    flow2, prior2, _, _ = build(3, 3, 1, 41)
    flow2.load_state_dict(sd); prior2.load_state_dict(psd)
    flow2.set_compute("bf16"); flow2.to(DEV); prior2.to(DEV)
    with torch.no_grad():
        ll, logp = initialize_with_zeros(2, 8, torch.device(DEV))
        latents, ll, logp = flow2.transform(x, ll, logp)
        fresh = float(calculate_loss(ll + logp + prior2.compute_log_prob(latents[-1]), 32.0, 32 * 32 * 3.0))
    assert after < before, (before, after)
    assert abs(after - fresh) < 1e-6 * abs(fresh), (after, fresh)


@pytest.mark.parametrize("shape,cin", [((4, 16, 16), 512), ((5, 11, 11), 128), ((2, 16, 16), 64)])
def test_pointwise_dgrad_with_fused_scale_relu_backward(shape, cin):
    """The data-gradient GEMM whose epilogue is the scale/ReLU backward of the layer in front: against the two-pass form
    (conv2d -> bf16 dy -> scale_bwd) on the same operands.  dc differs only by the bf16 rounding of dy the fused form skips;
    the parameter sums agree to that rounding."""
    b, h, w = shape
    cout = 512
    g = torch.Generator().manual_seed(cin)
    x = (torch.randn(b, h, w, cin, generator=g) * 0.5).bfloat16().to(DEV)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(DEV)
    y = torch.relu(torch.randn(b, h, w, cout, generator=g)).bfloat16().to(DEV)
    mul = (torch.randn(cout, generator=g) * 0.2).to(DEV)
    packed = K.conv_pack(wt, K.BF16)
    dy = K.conv2d(x, packed, cout, 1, compute=K.BF16, out_dtype=K.F32)                  # unrounded dy for the reference
    dc_ref, dmul_ref, dadd_ref = K.scale_bwd(dy, y.float(), mul, 1.0, True)
    dc, dmul, dadd = K.pointwise_dgrad_masked(x, packed, y, mul, 1.0)
    assert dc.dtype == torch.bfloat16
    assert_close(dc.float(), dc_ref, 4e-3, "dc")
    assert_close(dmul, dmul_ref, 2e-5, "dmul")
    assert_close(dadd, dadd_ref, 2e-5, "dadd")


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_parameter_gradients_added_straight_into_grad(compute):
    """training.DIRECT_PARAM_GRADS: with a dense .grad in place (FusedAdam's flat-buffer views) the finishing launches of the backward
    kernels add the parameter gradients there and the nodes return None - same numbers, bit for bit, as the gradients autograd
    accumulates from returned tensors (incl. the im2col-form weight gradients stored into OIHW by the reduce), a second backward
    doubles them, and a parameter with a post-accumulate hook (a gradient reducer) keeps the autograd route."""
    from normalizing_flow import calculate_loss, initialize_with_zeros, training
    from tests.golden.synth import DEEP
    B = 8
    flow, prior, _, _ = build(3, 3, 2, 41)
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 41, DEEP)) for k, v in flow.state_dict().items()})
    prior.load_state_dict({k: T(synth_tensor(k, v.shape, 42, DEEP)) for k, v in prior.state_dict().items()})
    flow.set_compute(compute)
    x = (T(synth_input("train_x", (B, 3, 32, 32), 6)) * 0.4).to(DEV)
    params = list(flow.parameters()) + list(prior.parameters())

    def backward():
        ll, logp = initialize_with_zeros(2, B, torch.device(DEV))
        latents, ll, logp = flow.transform(x, ll, logp)
        loss = calculate_loss(ll + logp + prior.compute_log_prob(latents[-1]), 32.0, 32 * 32 * 3.0)
        loss.backward()
        return loss.detach()

    backward()                                                        # data-dependent init out of the way; .grad = None so far
    training.enable_direct_grads(params)                              # opt-in: what FusedAdam(direct_grads=True) does
    def run(direct, times=1):
        training.DIRECT_PARAM_GRADS = direct
        for p in params:
            p.grad = torch.zeros_like(p)
        K.start_trace()
        for _ in range(times):
            backward()
        n_launch = len(K.stop_trace())
        return [p.grad.clone() for p in params], n_launch
    try:
        ref, _ = run(False)
        got, _ = run(True)
        for (name, _), a, b in zip(list(flow.named_parameters()) + list(prior.named_parameters()), got, ref):
            assert torch.equal(a, b), (name, float((a - b).abs().max()))
        twice, _ = run(True, times=2)
        for a, b in zip(twice, ref):
            assert_close(a, 2 * b, 1e-6, "two backward passes accumulate")
        seen = []
        w = flow.blocks[0].flows[0].affcoupling.net[2].conv.weight
        h = w.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.clone()))
        hooked, _ = run(True)
        h.remove()
        assert len(seen) == 1 and all(torch.equal(a, b) for a, b in zip(hooked, ref))
        # unmarked parameters (the default: an external reducer such as torch DDP may hang on AccumulateGrad) keep the autograd
        # route: every parameter's AccumulateGrad node runs
        training.enable_direct_grads(params, False)
        fired = []
        accs = [p.view_as(p).grad_fn.next_functions[0][0] for p in params]
        hs = [a.register_hook(lambda *_: fired.append(1)) for a in accs]   # node-level hooks, like DDP's (invisible to _slot)
        unmarked, _ = run(True)
        for h in hs:
            h.remove()
        assert len(fired) == len(params) and all(torch.equal(a, b) for a, b in zip(unmarked, ref))
    finally:
        training.DIRECT_PARAM_GRADS = True


def test_batched_weight_packing_matches_the_single_launches():
    """nfdp_conv_pack_weights_batch: all derived layouts of a step in one launch, byte for byte what conv_pack / conv_pack_ex give;
    a second run() picks up changed weights (the job table points at the parameters' storage)."""
    ws = [torch.nn.Parameter(rnd(512, 6, 3, 3, seed=1).to(DEV)), torch.nn.Parameter(rnd(512, 512, 1, 1, seed=2).to(DEV)),
          torch.nn.Parameter(rnd(12, 512, 3, 3, seed=3).to(DEV)), torch.nn.Parameter(rnd(48, 512, 3, 3, seed=4).to(DEV))]
    specs = [(ws[0], False, False), (ws[1], False, False), (ws[2], False, True), (ws[2], True, False), (ws[1], True, False),
             (ws[0], True, True), (ws[3], False, True), (ws[3], True, False)]
    for compute in (K.BF16, K.F32):
        pb = K.PackBatch(specs, compute)
        for rnd_no in range(2):
            pb.run()
            for (w, flip, taps), v in zip(specs, pb.views):
                ref = K.conv_pack_ex(w.detach(), compute, flip=flip, taps=taps)
                assert v.numel() == ref.numel() and torch.equal(v, ref), (tuple(w.shape), flip, taps, compute)
                if not flip and not taps:
                    assert torch.equal(v, K.conv_pack(w.detach(), compute))
            with torch.no_grad():
                for w in ws:
                    w.mul_(-0.5)
        assert pb.matches()
        ws[0].data = ws[0].data.clone()
        assert not pb.matches()


def test_training_step_at_the_benchmark_size():
    """BASELINE configs[2] shape on one GPU (Glow L3 K16, 32x32x3, B = 512, bf16): the training step at the size bench.py --train times -
    the chip-wide grids of the weight-gradient launches (40 / 24 / 24 pixel slices per level-0 flow), two workgroup rounds of the level-0
    kernels.  Size-independent properties: the training-path loss equals the inference-path bits/dim on the same weights (same kernels
    modulo the folded scales: bf16 round-off), every gradient is finite, a repeated step reproduces every gradient bit for bit, and the
    gradient of the batch is the mean of the gradients of its two halves (linearity of the batched sums, fp32 summation order aside)."""
    from normalizing_flow import calculate_loss, initialize_with_zeros
    from tests.golden.synth import DEEP
    B = 512
    flow, prior, _, _ = build(3, 3, 16, 71)
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 71, DEEP)) for k, v in flow.state_dict().items()})
    prior.load_state_dict({k: T(synth_tensor(k, v.shape, 72, DEEP)) for k, v in prior.state_dict().items()})
    flow.set_compute("bf16")
    x = (T(synth_input("train_x", (B, 3, 32, 32), 9)) * 0.4).to(DEV)
    params = list(flow.parameters()) + list(prior.parameters())

    def grads(xb):
        for p in params:
            p.grad = None
        n = xb.shape[0]
        ll, logp = initialize_with_zeros(2, n, torch.device(DEV))
        latents, ll, logp = flow.transform(xb, ll, logp)
        loss = calculate_loss(ll + logp + prior.compute_log_prob(latents[-1]), 32.0, 32 * 32 * 3.0)
        loss.backward()
        return loss.detach().clone(), [p.grad.clone() for p in params]
    l0, g0 = grads(x)
    l1, g1 = grads(x)
    assert all(torch.isfinite(g).all() for g in g0)
    assert torch.equal(l0, l1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
    with torch.no_grad():
        ll, logp = initialize_with_zeros(2, B, torch.device(DEV))
        latents, ll, logp = flow.transform(x, ll, logp)
        bpd = calculate_loss(ll + logp + prior.compute_log_prob(latents[-1]), 32.0, 32 * 32 * 3.0)
    assert_close(l0, bpd, 1e-4, "training-path loss vs inference-path bits/dim")
    la, ga = grads(x[: B // 2])
    lb, gb = grads(x[B // 2:])
    assert_close(0.5 * (la + lb), l0, 1e-5, "loss of the halves")
    fro = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    errs = [fro(0.5 * (a + b), c) for a, b, c in zip(ga, gb, g0) if float(c.abs().max()) > 0]
    assert max(errs) <= 2e-3, max(errs)


@pytest.mark.parametrize("B,direct,chan", [(8, False, 3), (3, True, 3), (37, False, 3), (5, False, 1), (16, True, 1)])
def test_fused_level_training_forward_matches_the_per_flow_nodes(B, direct, chan):
    """training.FUSED_TRAIN_FORWARD: the forward of every Glow level is ONE persistent launch (nfdp_level_fused_train_bf16) that
    writes out what the backward needs; the per-flow autograd nodes (im2col + three GEMMs + coupling per flow) are the yardstick.
    Same rounding points (bf16 MFMA inputs, fp32 accumulation and flow state), different summation order inside the GEMMs: latents,
    bits/dim and every parameter gradient agree to bf16 round-off; ragged batches exercise the padded rows of the saved hidden
    activations (8x8 / 4x4 levels own 2 / 8 images per workgroup).  chan = 1: the geometries of one-channel 32x32 images (C = 4 at
    16x16, 8 at 8x8, 16 at 4x4)."""
    from normalizing_flow import calculate_loss, initialize_with_zeros, training
    from tests.golden.synth import DEEP
    flow, prior, _, _ = build(chan, 3, 2, 51)
    flow.load_state_dict({k: T(synth_tensor(k, v.shape, 51, DEEP)) for k, v in flow.state_dict().items()})
    prior.load_state_dict({k: T(synth_tensor(k, v.shape, 52, DEEP)) for k, v in prior.state_dict().items()})
    flow.set_compute("bf16")
    x = (T(synth_input("train_x", (B, chan, 32, 32), 8)) * 0.4).to(DEV)
    params = list(flow.parameters()) + list(prior.parameters())
    names = [n for n, _ in flow.named_parameters()] + [n for n, _ in prior.named_parameters()]
    if direct:
        training.enable_direct_grads(params)

    def run(fused):
        training.FUSED_TRAIN_FORWARD = fused
        for p in params:
            p.grad = torch.zeros_like(p) if direct else None
        ll, logp = initialize_with_zeros(2, B, torch.device(DEV))
        K.start_trace()
        latents, ll, logp = flow.transform(x, ll, logp)
        loss = calculate_loss(ll + logp + prior.compute_log_prob(latents[-1]), 32.0, 32 * 32 * float(chan))
        loss.backward()
        labels = [e["label"] for e in K.stop_trace()]
        return [z.detach().clone() for z in latents], loss.detach().clone(), [p.grad.clone() for p in params], labels
    try:
        z0, l0, g0, lab0 = run(False)
        z1, l1, g1, lab1 = run(True)
        z2, l2, g2, _ = run(True)
    finally:
        training.FUSED_TRAIN_FORWARD = True
        training.enable_direct_grads(params, False)
    n_fused = 3
    assert sum(lb.startswith("level_fused_train") for lb in lab1) == n_fused and not any(lb.startswith("level_fused_train") for lb in lab0)
    assert len(lab1) < len(lab0)
    for a, b in zip(z1, z0):
        assert_close(a, b, 2e-2, "latents")                          # bf16 hidden activations: 8-bit mantissa
    assert_close(l1, l0, 1e-4, "bits/dim")
    fro = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    errs = {n: fro(a, b) for n, a, b in zip(names, g1, g0) if float(b.abs().max()) > 0}
    worst = max(errs.items(), key=lambda kv: kv[1])
    print("fused vs per-flow training forward: median rel Frobenius err", float(np.median(list(errs.values()))), "worst", worst)
    assert worst[1] <= 5e-2 and float(np.median(list(errs.values()))) <= 5e-3, worst
    assert torch.equal(l1, l2) and all(torch.equal(a, b) for a, b in zip(g1, g2))      # deterministic


@pytest.mark.parametrize("hw,c,B", [(16, 12, 3), (8, 24, 5), (4, 48, 11), (16, 12, 1), (16, 4, 2), (8, 8, 3), (4, 16, 9)])
def test_fused_reverse_coupling_network(hw, c, B):
    """nfdp_coupling_bwd_fused_bf16: the data-gradient chain of a coupling network (zero conv^T -> mask/scale -> 1x1^T -> mask/scale ->
    input conv^T) in one launch, against torch's transposed convolutions on the same bf16-rounded operands: the two hidden gradients
    (bf16, the forward's [pixel][512] layout, rows of absent images in the padded tail stay zero) and the input gradient added
    into dx[..., :C/2].  Ragged batches: the 8x8 / 4x4 kernels own 2 / 8 images per workgroup."""
    ch = c // 2
    bf = lambda t: t.to(torch.bfloat16).float()
    w1, w2, w3 = rnd(512, ch, 3, 3, seed=1, scale=0.05), rnd(512, 512, 1, 1, seed=2, scale=0.05), rnd(c, 512, 3, 3, seed=3, scale=0.05)
    s1, s2 = rnd(512, seed=4, scale=0.2), rnd(512, seed=5, scale=0.2)
    dc3 = rnd(B, c, hw, hw, seed=6)
    y1 = bf(torch.relu(rnd(B, 512, hw, hw, seed=7)))
    y2 = bf(torch.relu(rnd(B, 512, hw, hw, seed=8)))
    dx0 = rnd(B, c, hw, hw, seed=9)
    # reference (fp32 math on bf16-rounded GEMM operands, fp32 accumulation)
    dy2 = F.conv_transpose2d(bf(dc3), bf(w3), padding=1)
    dc2 = bf((y2 > 0) * dy2 * torch.exp(s2).view(1, -1, 1, 1))
    dy1 = F.conv_transpose2d(dc2, bf(w2))
    dc1 = bf((y1 > 0) * dy1 * torch.exp(s1).view(1, -1, 1, 1))
    dxa = F.conv_transpose2d(dc1, bf(w1), padding=1)
    grp = {16: 1, 8: 2, 4: 8}[hw]
    bpad = (B + grp - 1) // grp * grp
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    def padded(t):                                                 # saved activations: padded to whole workgroups, blocked layout
        out = torch.zeros(bpad, hw, hw, 512, dtype=torch.bfloat16)
        out[:B] = nhwc(t).to(torch.bfloat16)
        return K.rows_to_blocked(out).to(DEV)
    packed = K.coupling_bwd_fused_pack(w1.to(DEV), s1.to(DEV), w2.to(DEV), s2.to(DEV), w3.to(DEV))
    dx = nhwc(dx0).to(DEV)
    g1b, g2b, colb = K.coupling_bwd_fused(nhwc(dc3).to(DEV), dx, packed, padded(y1), padded(y2))
    assert g1b.dtype == torch.bfloat16 and tuple(g1b.shape) == (bpad, hw, hw, 512)
    g1, g2 = K.blocked_to_rows(g1b), K.blocked_to_rows(g2b)          # the gradients come back in the same blocked layout
    # im2col rows of dc3 (x operand of the zero conv's weight-gradient GEMM): the standalone kernel's columns, zero tail images
    ref_col = K.im2col3x3(nhwc(dc3).to(DEV))
    ncol = colb.shape[-1]
    assert ncol == 16 * ((9 * c + 15) // 16) and tuple(colb.shape[:3]) == (bpad, hw, hw)
    assert torch.equal(colb[:B, ..., : 9 * c], ref_col[..., : 9 * c]) and float(colb[:B, ..., 9 * c:].float().abs().max() if ncol > 9 * c else 0.0) == 0.0
    assert float(colb[B:].float().abs().max() if bpad > B else 0.0) == 0.0
    assert_close(g2[:B].float(), nhwc(dc2), 6e-2, "dc2")          # bf16 operands AND bf16 storage, element-wise: observed 2-4e-2
    assert_close(g1[:B].float(), nhwc(dc1), 6e-2, "dc1")
    want = nhwc(dx0).clone()
    want[..., :ch] += nhwc(dxa)
    assert_close(dx, want, 6e-2, "dx")
    assert torch.equal(dx[..., ch:].cpu(), nhwc(dx0)[..., ch:])
    assert float(g1[B:].float().abs().max() if bpad > B else 0.0) == 0.0 and float(g2[B:].float().abs().max() if bpad > B else 0.0) == 0.0
    # ReLU mask exactly: where the saved activation is zero the gradient is zero
    assert float(g2[:B].float()[nhwc(y2) == 0].abs().max()) == 0.0
    # determinism
    dx_b = nhwc(dx0).to(DEV)
    h1, h2, _ = K.coupling_bwd_fused(nhwc(dc3).to(DEV), dx_b, packed, padded(y1), padded(y2))
    assert torch.equal(h1, g1b) and torch.equal(h2, g2b) and torch.equal(dx_b, dx)


@pytest.mark.parametrize("hw,c,B", [(16, 12, 5), (8, 24, 7), (16, 4, 2), (8, 8, 3)])
def test_relu_masks_of_the_training_forward_drive_the_reverse_network(hw, c, B):
    """The training forward (nfdp_level_fused_train_ex_bf16) also writes the ReLU SIGNS of both hidden activations - one dword per lane
    and hidden block, 1/8 of the activations' bytes - and the reverse network reads them instead of the activations: (a) every mask bit
    equals [y > 0] of the saved activation at that (pixel, channel); (b) the reverse network's outputs (both hidden gradients, the im2col
    rows, dx) from the masks equal those from the activations BIT FOR BIT; (c) the 4x4 level has no masks and says so."""
    import numpy as np
    from normalizing_flow.glow import StepFlow
    from tests.golden.synth import synth_tensor, DEEP
    n = 2
    flows = []
    for i in range(n):
        f = StepFlow(c, 512, compute="bf16").to(DEV)
        f.load_state_dict({k: torch.from_numpy(np.array(synth_tensor(f"flows.{i}." + k, v.shape, 21 + i, DEEP))) for k, v in f.state_dict().items()})
        flows.append(f)
    assert K.lib().nfdp_level_fused_train_has_masks(hw, hw, c) == 1 and K.lib().nfdp_level_fused_train_has_masks(4, 4, 48) == 0
    x = (torch.randn(B, hw, hw, c, generator=torch.Generator().manual_seed(3)) * 0.4).to(DEV)
    ldj = torch.zeros(B, dtype=torch.float64, device=DEV)
    packed = [f.affcoupling.net.fused_packed_train() for f in flows]
    scales = [f.actnorm.scale.detach().view(-1) for f in flows]
    biases = [f.actnorm.bias.detach().view(-1) for f in flows]
    ws = [f.invconv2d._w() for f in flows]
    xpre, ypost, hout, y1, y2, xcol, masks = K.level_fused_train(x.clone(), packed, scales, biases, ws, ldj, want_masks=True)
    bpad = y1.shape[1]
    assert masks is not None and tuple(masks.shape) == (n, 2, bpad * hw * hw * 32) and masks.dtype == torch.int32
    # without masks the launch writes the same activations
    out2 = K.level_fused_train(x.clone(), packed, scales, biases, ws, torch.zeros_like(ldj), want_masks=False)
    assert torch.equal(out2[3], y1) and torch.equal(out2[4], y2) and len(out2) == 6
    # (a) decode: tile t, hidden block hb, lane l, entry e = 4 s + d: low half (value 2d) at bit 7 - e, high half (2d + 1) at bit 23 - e;
    #     the blocked activation tensor holds the same lane's 8 bf16 of k-step s at [t][hb][s][l][0..7]
    for i in range(n):
        for layer, y in ((0, y1[i]), (1, y2[i])):
            m = masks[i, layer].view(-1, 16, 64).cpu().numpy().astype(np.uint32)            # [tile][hb][lane]
            yb = y.reshape(-1).view(torch.int16).cpu().numpy().reshape(-1, 16, 2, 64, 8)     # [tile][hb][s][lane][j]
            for s_ in range(2):
                for d in range(4):
                    e = 4 * s_ + d
                    lo = (m >> (7 - e)) & 1
                    hi = (m >> (23 - e)) & 1
                    assert np.array_equal(lo, (yb[:, :, s_, :, 2 * d] != 0).astype(np.uint32)), (i, layer, s_, d)
                    assert np.array_equal(hi, (yb[:, :, s_, :, 2 * d + 1] != 0).astype(np.uint32)), (i, layer, s_, d)
    # (b) the reverse network from masks vs from activations
    net = flows[1].affcoupling.net
    rev = K.coupling_bwd_fused_pack(net[0].conv.weight.detach(), net[0].actnorm.scale.detach().view(-1), net[2].conv.weight.detach(),
                                    net[2].actnorm.scale.detach().view(-1), net[4].weight.detach())
    dc3 = (torch.randn(B, hw, hw, c, generator=torch.Generator().manual_seed(4)) * 0.3).to(DEV)
    dx_a = torch.randn(B, hw, hw, c, generator=torch.Generator().manual_seed(5)).to(DEV)
    dx_b = dx_a.clone()
    a1, a2, acol = K.coupling_bwd_fused(dc3, dx_a, rev, y1[1], y2[1])
    b1, b2, bcol = K.coupling_bwd_fused(dc3, dx_b, rev, None, None, masks=masks[1])
    assert torch.equal(a1, b1) and torch.equal(a2, b2) and torch.equal(acol, bcol) and torch.equal(dx_a, dx_b)
    assert float(a2.float().abs().max()) > 0
    with pytest.raises(K.NfdpError):
        K.coupling_bwd_fused(torch.zeros(8, 4, 4, 48, device=DEV), torch.zeros(8, 4, 4, 48, device=DEV), rev, None, None,
                             masks=torch.zeros(2, 8 * 16 * 32, dtype=torch.int32, device=DEV))


@pytest.mark.parametrize("npix,cin,cout,form_c", [(1024, 512, 512, 0), (640, 64, 512, 6), (2048, 128, 512, 12), (128, 512, 512, 0)])
def test_weight_gradient_gemm_with_actnorm_gradients(npix, cin, cout, form_c):
    """nfdp_conv_wgrad_bf16_affine: dW of a Conv2dActNorm layer and, from the same GEMM, its ActNorm gradients - dbias as column sums
    of the raw-output gradient (ones-fragment MFMAs), dscale through  sum_p dy*y = sum_n W[m][n] dW[m][n] + b[m] db[m]  - against
    autograd of y = relu((x W^T + b) e^s) on the same bf16 operands; accumulation into existing gradients; im2col form (form_c > 0:
    x rows are (tap, channel) columns padded to 64 / 128, dW stored as [Cout, form_c, 3, 3])."""
    bf = lambda t: t.to(torch.bfloat16).float()
    k_real = 9 * form_c if form_c else cin
    x = bf(rnd(npix, cin, seed=1))
    if form_c:
        x[:, k_real:] = 0.0
    w = rnd(cout, k_real, seed=2, scale=0.05)
    b = rnd(cout, seed=3, scale=0.3)
    s_ = rnd(cout, seed=4, scale=0.2)
    wq = bf(w).requires_grad_()
    bq, sq = b.clone().requires_grad_(), s_.clone().requires_grad_()
    u = x[:, :k_real] @ wq.t()
    y = torch.relu((u + bq) * torch.exp(sq))
    dy = rnd(npix, cout, seed=5)
    dc = bf((y > 0) * dy * torch.exp(sq).detach())                         # what the fused backward kernel writes (bf16)
    # autograd with the bf16-rounded dc as the upstream gradient at the raw conv output
    (u * dc).sum().backward(inputs=[wq], retain_graph=True)
    dw_ref = wq.grad
    db_ref = dc.sum(0)
    ds_ref = (dc * (u.detach() + b)).sum(0)
    if form_c:
        w_param = w.view(cout, 3, 3, form_c).permute(0, 3, 1, 2).contiguous()      # columns are (tap, channel); parameter is OIHW
        dw_want = dw_ref.view(cout, 3, 3, form_c).permute(0, 3, 1, 2).contiguous()
    else:
        w_param, dw_want = w.view(cout, cin, 1, 1), dw_ref.view(cout, cin, 1, 1)
    dw = torch.full(w_param.shape, 0.5, device=DEV)
    dsc, dbi = torch.full((cout,), -1.0, device=DEV), torch.full((cout,), 2.0, device=DEV)
    args = (x.to(torch.bfloat16).to(DEV), dc.to(torch.bfloat16).to(DEV), bf(w_param).to(DEV).contiguous(), b.to(DEV))
    K.conv_wgrad_affine(*args, dw, dsc, dbi, accumulate=True, form=1 if form_c else 0, form_c=form_c)
    assert_close(dw - 0.5, dw_want, 2e-3, "dW (accumulated)")
    assert_close(dbi - 2.0, db_ref, 2e-3, "dbias")
    assert_close(dsc + 1.0, ds_ref, 5e-3, "dscale")
    dw2, dsc2, dbi2 = torch.empty_like(dw), torch.empty_like(dsc), torch.empty_like(dbi)
    K.conv_wgrad_affine(*args, dw2, dsc2, dbi2, accumulate=False, form=1 if form_c else 0, form_c=form_c)
    assert_close(dw2, dw_want, 2e-3, "dW")
    assert_close(dsc2, ds_ref, 5e-3, "dscale (stored)")
    dw3, dsc3, dbi3 = torch.empty_like(dw), torch.empty_like(dsc), torch.empty_like(dbi)
    K.conv_wgrad_affine(*args, dw3, dsc3, dbi3, accumulate=False, form=1 if form_c else 0, form_c=form_c)
    assert torch.equal(dw2, dw3) and torch.equal(dsc2, dsc3) and torch.equal(dbi2, dbi3)
    # 512-channel operands in the fused kernels' blocked layout: same numbers, bit for bit (only the source addresses differ)
    xb = K.rows_to_blocked(args[0]) if cin == 512 else args[0]
    dyb = K.rows_to_blocked(args[1])
    dw4, dsc4, dbi4 = torch.empty_like(dw), torch.empty_like(dsc), torch.empty_like(dbi)
    K.conv_wgrad_affine(xb, dyb, args[2], args[3], dw4, dsc4, dbi4, accumulate=False, form=1 if form_c else 0, form_c=form_c,
                        x_blocked=cin == 512, dy_blocked=True)
    assert torch.equal(dw2, dw4) and torch.equal(dsc2, dsc4) and torch.equal(dbi2, dbi4)
    if not form_c:
        dw5 = torch.empty_like(dw)
        K.conv_wgrad_blocked(xb, dyb, dw5, False, x_blocked=True, dy_blocked=True)
        assert_close(dw5, dw_want, 2e-3, "plain dW on blocked operands")


@pytest.mark.parametrize("npix,c", [(2048, 12), (512, 48), (96, 24), (8192, 4)])
def test_weight_gradients_of_a_flow_in_one_launch(npix, c):
    """nfdp_coupling_wgrad3_bf16: the three weight gradients of a coupling network (512 x 512 on blocked operands with ActNorm
    gradients, the input conv's on im2col rows with ActNorm gradients, the zero conv's in the transposed form) as one GEMM launch and
    one reduce launch, against the per-gradient entry points on the same operands (different pixel slices: fp32 summation order
    differs), store and accumulate modes, and with one or two problems only."""
    bf = lambda t: t.to(torch.bfloat16)
    ch = c // 2
    k1, k3 = 16 * ((9 * ch + 15) // 16), 16 * ((9 * c + 15) // 16)
    y1 = K.rows_to_blocked(bf(torch.relu(rnd(npix, 512, seed=1))).to(DEV))
    y2 = K.rows_to_blocked(bf(torch.relu(rnd(npix, 512, seed=2))).to(DEV))
    dc1 = K.rows_to_blocked(bf(rnd(npix, 512, seed=3)).to(DEV))
    dc2 = K.rows_to_blocked(bf(rnd(npix, 512, seed=4)).to(DEV))
    xcol = rnd(npix, k1, seed=5); xcol[:, 9 * ch:] = 0.0
    dcol = rnd(npix, k3, seed=6); dcol[:, 9 * c:] = 0.0
    xcol, dcol = bf(xcol).to(DEV), bf(dcol).to(DEV)
    w1, b1 = rnd(512, ch, 3, 3, seed=7, scale=0.05).to(DEV), rnd(512, seed=8, scale=0.3).to(DEV)
    w2, b2 = rnd(512, 512, 1, 1, seed=9, scale=0.05).to(DEV), rnd(512, seed=10, scale=0.3).to(DEV)
    z = lambda *shape: torch.zeros(*shape, device=DEV)
    # the per-gradient entry points
    gw1, gs1, gb1, gw2, gs2, gb2, gw3 = z(512, ch, 3, 3), z(512), z(512), z(512, 512, 1, 1), z(512), z(512), z(c, 512, 3, 3)
    K.conv_wgrad_affine(y1, dc2, w2, b2, gw2, gs2, gb2, False, x_blocked=True, dy_blocked=True)
    K.conv_wgrad_affine(xcol, dc1, w1, b1, gw1, gs1, gb1, False, form=1, form_c=ch, dy_blocked=True)
    K.conv_wgrad_blocked(dcol, y2, gw3, False, form=2, form_c=c, dy_blocked=True)

    def problems(hw1, hs1, hb1, hw2, hs2, hb2, hw3):
        return [dict(x=y1, dy=dc2, dw=hw2, x_blocked=True, weight=w2, bias=b2, dscale=hs2, dbias=hb2),
                dict(x=xcol, dy=dc1, dw=hw1, form=1, form_c=ch, weight=w1, bias=b1, dscale=hs1, dbias=hb1),
                dict(x=dcol, dy=y2, dw=hw3, form=2, form_c=c)]
    out = [torch.full_like(t, 0.25) for t in (gw1, gs1, gb1, gw2, gs2, gb2, gw3)]
    K.coupling_wgrad3(problems(*out), accumulate=True)
    for got, want, what in zip(out, (gw1, gs1, gb1, gw2, gs2, gb2, gw3), ("dW1", "dscale1", "dbias1", "dW2", "dscale2", "dbias2", "dW3")):
        assert_close(got - 0.25, want, 2e-4, f"{what} (accumulated)")
    out2 = [torch.empty_like(t) for t in out]
    K.coupling_wgrad3(problems(*out2), accumulate=False)
    out3 = [torch.empty_like(t) for t in out]
    K.coupling_wgrad3(problems(*out3), accumulate=False)
    for a, b_, want in zip(out2, out3, (gw1, gs1, gb1, gw2, gs2, gb2, gw3)):
        assert torch.equal(a, b_)                                               # fixed summation order: run-to-run identical
        assert_close(a, want, 2e-4, "stored")
    one = torch.empty_like(gw3)
    K.coupling_wgrad3([dict(x=dcol, dy=y2, dw=one, form=2, form_c=c)], accumulate=False)
    assert_close(one, gw3, 2e-4, "a single problem")
    two = [torch.empty_like(gw2), torch.empty_like(gs2), torch.empty_like(gb2), torch.empty_like(gw3)]
    K.coupling_wgrad3([dict(x=y1, dy=dc2, dw=two[0], x_blocked=True, weight=w2, bias=b2, dscale=two[1], dbias=two[2]),
                       dict(x=dcol, dy=y2, dw=two[3], form=2, form_c=c)], accumulate=False)
    for got, want in zip(two, (gw2, gs2, gb2, gw3)):
        assert_close(got, want, 2e-4, "two problems")
    with pytest.raises(Exception):
        K.coupling_wgrad3([dict(x=xcol[:-1], dy=dc1, dw=gw1, form=1, form_c=ch)])


@pytest.mark.parametrize("c,hw,B", [(12, 16, 3), (24, 8, 5), (48, 4, 21), (4, 16, 2), (16, 4, 7)])
def test_flow_link_backward_kernel(c, hw, B):
    """nfdp_flow_link_bwd_f32 + nfdp_flow_link_finish_f32: flow head backwards + previous flow's coupling backwards + zero-conv scale
    backward in one launch, against the per-op kernels it replaces (ActNorm recompute, W^T product, two scale backwards, coupling
    backward): all three call shapes of a level (coupling only / both / head only), accumulate and store modes."""
    ch = c // 2
    T4 = lambda seed, s=1.0: rnd(B, hw, hw, c, seed=seed, scale=s).to(DEV)
    dx_in, xpre, ypost, hout = T4(1), T4(2), T4(3), T4(4, 0.5)
    w = (torch.linalg.qr(rnd(c, c, seed=5))[0] + 0.05 * rnd(c, c, seed=6)).contiguous().to(DEV)
    scale, bias, l3 = rnd(c, seed=7, scale=0.2).to(DEV), rnd(c, seed=8, scale=0.3).to(DEV), rnd(c, seed=9, scale=0.1).to(DEV)
    gl = rnd(B, seed=10).double().to(DEV)
    f3 = 3.0
    npix = B * hw * hw
    # per-op reference
    a = K.actnorm_fwd(xpre, scale, bias)
    da = K.invconv_apply(dx_in, w.t().contiguous())
    dy, dscale, dbias = K.scale_bwd(da, a, scale, 1.0, False)
    dxo, dh = K.coupling_bwd(ypost, hout, dy, gl)
    dc3, dl3, db3 = K.scale_bwd(dh, hout, l3, f3, False)
    dxo_c, dh_c = K.coupling_bwd(ypost, hout, dx_in, gl)             # coupling only: dy = dx_in
    dc3_c, dl3_c, db3_c = K.scale_bwd(dh_c, hout, l3, f3, False)
    part = K.flow_link_part(3, npix, c, DEV)
    o0 = K.flow_link_bwd(dx_in, part[0], coupling=(ypost, hout, gl, l3, f3))
    o1 = K.flow_link_bwd(dx_in, part[1], head=(xpre, w, scale, bias), coupling=(ypost, hout, gl, l3, f3))
    o2 = K.flow_link_bwd(dx_in, part[2], head=(xpre, w, scale, bias))
    assert o0[2] is None and o2[1] is None
    assert_close(o0[0], dxo_c, 1e-6, "coupling only: dx"); assert_close(o0[1], dc3_c, 1e-6, "coupling only: dc3")
    assert_close(o1[0], dxo, 1e-5, "both: dx"); assert_close(o1[1], dc3, 1e-5, "both: dc3"); assert_close(o1[2], a, 1e-6, "both: a")
    assert_close(o2[0], dy, 1e-5, "head only: dy"); assert_close(o2[2], a, 1e-6, "head only: a")
    z = lambda v=0.0: torch.full((c,), v, device=DEV)
    ds_, db_, dl_, d3_ = [None, z(), z()], [None, z(), z()], [z(), z(), None], [z(), z(), None]
    K.flow_link_finish(part, npix, c, f3, ds_, db_, dl_, d3_, accumulate=False)
    assert_close(dl_[0], dl3_c, 2e-5, "dlogs3 (coupling only)"); assert_close(d3_[0], db3_c, 2e-5, "dbias3 (coupling only)")
    assert_close(ds_[1], dscale, 2e-5, "dscale"); assert_close(db_[1], dbias, 2e-5, "dbias")
    assert_close(dl_[1], dl3, 2e-5, "dlogs3"); assert_close(d3_[1], db3, 2e-5, "dbias3")
    assert_close(ds_[2], dscale, 2e-5, "dscale (head only)")
    acc = [z(1.0), z(2.0)]
    K.flow_link_finish(part, npix, c, f3, [None, acc[0], None], [None, None, None], [None, None, None], [None, acc[1], None], accumulate=True)
    assert_close(acc[0] - 1.0, dscale, 2e-5, "accumulated dscale"); assert_close(acc[1] - 2.0, db3, 2e-5, "accumulated dbias3")
    # the 1x1 conv's weight gradient dW[k][c] = sum_p dx_in[p][k] a[p][c] (transforms.py:132) out of the head calls' partial buffers
    dw_ref = (dx_in.reshape(npix, c).double().t() @ a.reshape(npix, c).double()).float()
    dws = [None, torch.empty(c, c, 1, 1, device=DEV), torch.full((c, c), 0.5, device=DEV)]
    none3 = [None, None, None]
    K.flow_link_finish(part, npix, c, f3, none3, none3, none3, none3, accumulate=False, dw=[None, dws[1], None])
    K.flow_link_finish(part, npix, c, f3, none3, none3, none3, none3, accumulate=True, dw=[None, None, dws[2]])
    assert_close(dws[1].view(c, c), dw_ref, 1e-5, "1x1 conv dW (both)")
    assert_close(dws[2] - 0.5, dw_ref, 1e-5, "1x1 conv dW (head only, accumulated)")
    assert_close(K.conv_wgrad(a, dx_in, 1).view(c, c), dw_ref, 1e-5, "the split-K GEMM it replaces")


def test_activation_recomputation_gives_bit_identical_gradients():
    """training.RECOMPUTE_FLOWS = G: the fused training forward keeps only the flow-state tensors and the backward pass re-runs it per
    group of G flows from the group's saved input - the same kernel on the same fp32 inputs, so loss and every parameter gradient
    must equal the store-everything step bit for bit, for a group size that divides the level and one that does not."""
    from normalizing_flow import Glow, GaussianPrior, calculate_loss, training
    B = 6
    x = (T(synth_input("recomp_x", (B, 3, 32, 32), 9)) * 0.4).to(DEV)

    def run(group):
        flow, prior = Glow(in_channel=3, L=3, K=4, compute="bf16").to(DEV), GaussianPrior(48).to(DEV)
        flow.load_state_dict({k: T(synth_tensor(k, v.shape, 5, DEEP)) for k, v in flow.state_dict().items()})
        prior.load_state_dict({k: T(synth_tensor(k, v.shape, 5, DEEP)) for k, v in prior.state_dict().items()})
        training.RECOMPUTE_FLOWS = group
        try:
            K.start_trace()
            ldj = torch.zeros(B, dtype=torch.float64, device=DEV)
            logp = torch.zeros(B, dtype=torch.float64, device=DEV)
            zs, ldj, logp = flow.transform(x, ldj, logp)
            loss = calculate_loss(ldj + logp + prior.compute_log_prob(zs[-1]), 32.0, 3072.0)
            loss.backward()
            labels = [e["label"] for e in K.stop_trace()]
        finally:
            training.RECOMPUTE_FLOWS = 0
        n_fwd = sum(lb.startswith("level_fused_train") for lb in labels)
        return loss.detach(), {k: p.grad.clone() for k, p in flow.named_parameters() if p.grad is not None}, n_fwd

    l0, g0, n0 = run(0)
    assert n0 == 3                                                   # one training-forward launch per level
    for group in (2, 3):
        l1, g1, n1 = run(group)
        assert n1 == 3 + 3 * ((4 + group - 1) // group)              # + one launch per group and level in the backward pass
        assert torch.equal(l0, l1)
        assert g0.keys() == g1.keys()
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k
    # per level by spatial size: only the 16x16 level (which holds most of the memory) recomputes, in groups of 2
    l2, g2, n2 = run({16: 2})
    assert n2 == 3 + 2
    assert torch.equal(l0, l2) and all(torch.equal(g0[k], g2[k]) for k in g0)


def test_two_graph_training_step_repacks_its_weights():
    """The data-parallel form of the captured training step (bench.py --train --graph-multi): forward + backward as one hipGraph, the
    optimizer as another, an eager gradient exchange between the replays.  The forward graph must have been captured from the state
    every replay starts from (right after an optimizer update: weight-layout caches stale), so that it re-packs the weights the
    previous replay updated - captured after a warm-up without an optimizer step it would replay on the weights of capture time.
    Checked: three replayed steps equal three eager steps (loss and every parameter)."""
    from normalizing_flow import Glow, GaussianPrior, calculate_loss
    from nfdp_hip.optim import FusedAdam
    from nfdp_hip.graphs import capture_step
    B = 8
    x = (T(synth_input("two_graph_x", (B, 3, 32, 32), 4)) * 0.4).to(DEV)

    def make():
        flow, prior = Glow(in_channel=3, L=3, K=2, compute="bf16").to(DEV), GaussianPrior(48).to(DEV)
        flow.load_state_dict({k: T(synth_tensor(k, v.shape, 6, DEEP)) for k, v in flow.state_dict().items()})
        prior.load_state_dict({k: T(synth_tensor(k, v.shape, 6, DEEP)) for k, v in prior.state_dict().items()})
        opt = FusedAdam(clipped=flow.parameters(), unclipped=prior.parameters(), lr=1e-4, clip_value=1, max_norm=1, direct_grads=True)

        def fwd_bwd():
            ldj = torch.zeros(B, dtype=torch.float64, device=DEV)
            logp = torch.zeros(B, dtype=torch.float64, device=DEV)
            zs, ldj, logp = flow.transform(x, ldj, logp)
            loss = calculate_loss(ldj + logp + prior.compute_log_prob(zs[-1]), 32.0, 3072.0)
            opt.zero_grad()
            loss.backward()
            return loss.detach()

        def step():
            loss = fwd_bwd()
            opt.step()
            return loss
        return opt, fwd_bwd, step

    oe, _, eager = make()
    oc, fwd_bwd, cstep = make()
    g_fb = capture_step(fwd_bwd, warmup=1, mutates_params=False, warmup_fn=cstep)
    g_opt = capture_step(oc.step, warmup=1, warmup_fn=cstep)
    eager(); eager()                                                  # the two warm-up steps of the captured copy
    for k in range(3):
        l_e = eager()
        l_c = g_fb()
        g_opt()                                                       # (a multi-rank run all-reduces the flat gradient buffer between the two)
        torch.cuda.synchronize()
        assert_close(l_c, l_e, 1e-6, f"step {k}: loss")
        assert relerr(oc.flat_param, oe.flat_param) <= 1e-6, f"step {k}: parameters {relerr(oc.flat_param, oe.flat_param):.2e}"
    assert float((oe.flat_param - oc.flat_param).abs().max()) < 1e-5


@pytest.mark.parametrize("c,hw,B", [(24, 8, 5), (48, 4, 21), (48, 4, 512), (24, 8, 64)])
def test_link_kernel_with_several_threads_per_pixel_is_bit_identical(c, hw, B):
    """flow_link_bwd_split_kernel (C = 24: two, C = 48: four threads per pixel; a wave per MFMA tile; the four column sums side by side)
    against flow_link_bwd_kernel<C, 64> (one thread per pixel): every output tensor and every partial (fp64 column sums, the 1x1 conv's
    weight-gradient tiles) bit for bit, for the three call shapes of a level, ragged last workgroups included."""
    T4 = lambda seed, s=1.0: rnd(B, hw, hw, c, seed=seed, scale=s).to(DEV)
    dx_in, xpre, ypost, hout = T4(1), T4(2), T4(3), T4(4, 0.5)
    w = (torch.linalg.qr(rnd(c, c, seed=5))[0] + 0.05 * rnd(c, c, seed=6)).contiguous().to(DEV)
    scale, bias, l3 = rnd(c, seed=7, scale=0.2).to(DEV), rnd(c, seed=8, scale=0.3).to(DEV), rnd(c, seed=9, scale=0.1).to(DEV)
    gl = rnd(B, seed=10).double().to(DEV)
    npix = B * hw * hw
    setter = K.lib().nfdp_debug_set_link_split

    def run():
        part = K.flow_link_part(3, npix, c, DEV)
        part.zero_()
        outs = [K.flow_link_bwd(dx_in, part[0], coupling=(ypost, hout, gl, l3, 3.0)),
                K.flow_link_bwd(dx_in, part[1], head=(xpre, w, scale, bias), coupling=(ypost, hout, gl, l3, 3.0)),
                K.flow_link_bwd(dx_in, part[2], head=(xpre, w, scale, bias))]
        return outs, part

    new, pn = run()
    setter(0)
    try:
        old, po = run()
    finally:
        setter(1)
    for a, b in zip(new, old):
        for x, y in zip(a, b):
            assert (x is None) == (y is None)
            if x is not None:
                assert torch.equal(x, y)
    assert torch.equal(pn.view(torch.int64), po.view(torch.int64))
