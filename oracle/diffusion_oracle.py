"""CPU ORACLE (test infrastructure, NOT product code) for the diffusion-prior hot path.

Functional torch-CPU fp32 restatement of diffusion_prior/gaussian_diffusion.py: beta schedules and
the 13 derived buffers, the U-Net eps-predictor forward over a flat state-dict, q_sample /
p_sample / p_sample_loop / ddim_sample / p_losses with all randomness injected.  Each function
cites the reference file:line it follows.

Parity pin: golden vectors produced by importing the reference (tools/make_goldens.py ->
tests/golden/diffusion*.npz), checked in tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------- schedules
def betas_for(schedule: str, T: int, **kw) -> Tensor:
    """fp64 betas  (gaussian_diffusion.py:501-537)."""
    if schedule == "linear":
        sc = 1000 / T
        return torch.linspace(sc * 0.0001, sc * 0.02, T, dtype=torch.float64)
    t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
    if schedule == "cosine":
        s = kw.get("s", 0.008)
        ac = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    elif schedule == "sigmoid":
        start, end, tau = kw.get("start", -3), kw.get("end", 3), kw.get("tau", 1)
        v0 = torch.tensor(start / tau).sigmoid()
        v1 = torch.tensor(end / tau).sigmoid()
        ac = (-((t * (end - start) + start) / tau).sigmoid() + v1) / (v1 - v0)
    else:
        raise ValueError(f"unknown beta schedule {schedule}")
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
    "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
    "p2_loss_weight",
)


def diffusion_buffers(schedule: str, T: int, p2_gamma: float = 0.0, p2_k: float = 1.0, **kw) -> Dict[str, Tensor]:
    """The 13 fp32 [T] buffers, computed in fp64 then rounded  (gaussian_diffusion.py:582-634)."""
    betas = betas_for(schedule, T, **kw)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    b = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": torch.log(pv.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "p2_loss_weight": (p2_k + ac / (1 - ac)) ** -p2_gamma,
    }
    return {k: v.to(torch.float32) for k, v in b.items()}


def _ext(a: Tensor, t: Tensor, ndim: int) -> Tensor:
    """a[t] broadcast to [B,1,1,1]  (gaussian_diffusion.py:495-498)."""
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


# ----------------------------------------------------------------------------- U-Net pieces
def ws_conv(x: Tensor, w: Tensor, b: Tensor, padding: int) -> Tensor:
    """Weight-standardised conv: biased var, eps 1e-5 (fp32)  (gaussian_diffusion.py:189-197)."""
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return F.conv2d(x, (w - mean) * (var + 1e-5).rsqrt(), b, padding=padding)


def chan_layernorm(x: Tensor, g: Tensor) -> Tensor:
    """Channel LayerNorm with gain only  (gaussian_diffusion.py:205-209)."""
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def block(x: Tensor, sd: SD, p: str, groups: int, scale_shift=None) -> Tensor:
    """WS-conv3x3 -> GroupNorm -> x*(scale+1)+shift -> SiLU  (gaussian_diffusion.py:267-276)."""
    x = ws_conv(x, sd[p + "proj.weight"], sd[p + "proj.bias"], 1)
    x = F.group_norm(x, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-5)
    if scale_shift is not None:
        sc, sh = scale_shift
        x = x * (sc + 1) + sh
    return F.silu(x)


def resnet_block(x: Tensor, temb: Tensor, sd: SD, p: str, groups: int) -> Tensor:
    """(gaussian_diffusion.py:290-302)."""
    te = F.linear(F.silu(temb), sd[p + "mlp.1.weight"], sd[p + "mlp.1.bias"])[:, :, None, None]
    h = block(x, sd, p + "block1.", groups, te.chunk(2, dim=1))
    h = block(h, sd, p + "block2.", groups)
    if (p + "res_conv.weight") in sd:
        x = F.conv2d(x, sd[p + "res_conv.weight"], sd[p + "res_conv.bias"])
    return h + x


def linear_attention(x: Tensor, sd: SD, p: str, heads: int = 4) -> Tensor:
    """q softmax over d, k softmax over n, v/(hw); context = k v^T; out = context^T q  (gaussian_diffusion.py:318-333)."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, sd[p + "to_qkv.weight"])
    d = qkv.shape[1] // (3 * heads)
    q, k, v = (t.reshape(b, heads, d, h * w) for t in qkv.chunk(3, dim=1))
    q = q.softmax(dim=-2) * (d ** -0.5)
    k = k.softmax(dim=-1)
    v = v / (h * w)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, heads * d, h, w)
    out = F.conv2d(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return chan_layernorm(out, sd[p + "to_out.1.g"])


def full_attention(x: Tensor, sd: SD, p: str, heads: int = 4) -> Tensor:
    """softmax(q^T k * d^-0.5) v  (gaussian_diffusion.py:347-358)."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, sd[p + "to_qkv.weight"])
    d = qkv.shape[1] // (3 * heads)
    q, k, v = (t.reshape(b, heads, d, h * w) for t in qkv.chunk(3, dim=1))
    sim = torch.einsum("bhdi,bhdj->bhij", q * (d ** -0.5), k)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)          # b h n d
    out = out.permute(0, 1, 3, 2).reshape(b, heads * d, h, w)
    return F.conv2d(out, sd[p + "to_out.weight"], sd[p + "to_out.bias"])


def time_embedding(t: Tensor, sd: SD, dim: int) -> Tensor:
    """sinusoidal(dim) -> Linear -> GELU(exact) -> Linear  (gaussian_diffusion.py:231-237, 407-412)."""
    if "time_mlp.0.weights" in sd:            # RandomOrLearnedSinusoidalPosEmb (gaussian_diffusion.py:240-255)
        x = t[:, None]
        freqs = x * sd["time_mlp.0.weights"][None, :] * 2 * math.pi
        e = torch.cat((x, torch.cat((freqs.sin(), freqs.cos()), dim=-1)), dim=-1)
        e = F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
        return F.linear(F.gelu(e), sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])
    half = dim // 2
    f = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half) * -f)
    e = t[:, None] * f[None, :]
    e = torch.cat((e.sin(), e.cos()), dim=-1)
    e = F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    return F.linear(F.gelu(e), sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


def _squeeze2(x: Tensor) -> Tensor:
    b, c, hh, ww = x.shape
    return x.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, 4 * c, hh // 2, ww // 2)


def unet_forward(x: Tensor, time: Tensor, sd: SD, dim: int, dim_mults: Sequence[int], groups: int = 8,
                 self_condition: bool = False, x_self_cond: Optional[Tensor] = None) -> Tensor:
    """eps-hat = Unet(x_t, t [, x_self_cond])  (gaussian_diffusion.py:451-490, ctor 364-449); self_condition: the input is
    cat((x_self_cond or zeros, x), dim=1)  (:452-454)."""
    n_res = len(dim_mults)
    if self_condition:
        x = torch.cat((torch.zeros_like(x) if x_self_cond is None else x_self_cond, x), dim=1)
    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    r = x
    temb = time_embedding(time, sd, dim)
    hs = []
    for i in range(n_res):
        p = f"downs.{i}."
        x = resnet_block(x, temb, sd, p + "0.", groups)
        hs.append(x)
        x = resnet_block(x, temb, sd, p + "1.", groups)
        x = linear_attention(chan_layernorm(x, sd[p + "2.fn.norm.g"]), sd, p + "2.fn.fn.") + x
        hs.append(x)
        if i < n_res - 1:   # Downsample = squeeze + 1x1   (:176-180)
            x = F.conv2d(_squeeze2(x), sd[p + "3.1.weight"], sd[p + "3.1.bias"])
        else:
            x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
    x = resnet_block(x, temb, sd, "mid_block1.", groups)
    x = full_attention(chan_layernorm(x, sd["mid_attn.fn.norm.g"]), sd, "mid_attn.fn.fn.") + x
    x = resnet_block(x, temb, sd, "mid_block2.", groups)
    for i in range(n_res):
        p = f"ups.{i}."
        x = resnet_block(torch.cat((x, hs.pop()), dim=1), temb, sd, p + "0.", groups)
        x = resnet_block(torch.cat((x, hs.pop()), dim=1), temb, sd, p + "1.", groups)
        x = linear_attention(chan_layernorm(x, sd[p + "2.fn.norm.g"]), sd, p + "2.fn.fn.") + x
        if i < n_res - 1:   # Upsample = nearest x2 + 3x3  (:169-173)
            x = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), sd[p + "3.1.weight"], sd[p + "3.1.bias"], padding=1)
        else:
            x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
    x = resnet_block(torch.cat((x, r), dim=1), temb, sd, "final_res_block.", groups)
    return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])


# ----------------------------------------------------------------------------- diffusion process
def q_sample(buf, x0: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """sqrt(ac_t) x0 + sqrt(1-ac_t) eps  (gaussian_diffusion.py:806-812)."""
    return _ext(buf["sqrt_alphas_cumprod"], t, x0.ndim) * x0 + _ext(buf["sqrt_one_minus_alphas_cumprod"], t, x0.ndim) * noise


def predict_start_from_noise(buf, x_t: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """(gaussian_diffusion.py:641-645)."""
    return _ext(buf["sqrt_recip_alphas_cumprod"], t, x_t.ndim) * x_t - _ext(buf["sqrt_recipm1_alphas_cumprod"], t, x_t.ndim) * noise


def q_posterior(buf, x0: Tensor, x_t: Tensor, t: Tensor):
    """(gaussian_diffusion.py:665-672)."""
    mean = _ext(buf["posterior_mean_coef1"], t, x_t.ndim) * x0 + _ext(buf["posterior_mean_coef2"], t, x_t.ndim) * x_t
    return mean, _ext(buf["posterior_variance"], t, x_t.ndim), _ext(buf["posterior_log_variance_clipped"], t, x_t.ndim)


def p_sample(buf, eps_model: Callable[[Tensor, Tensor], Tensor], x: Tensor, t: int, noise: Optional[Tensor]):
    """One reverse step, x0 clamped to [-1,1]; noise ignored at t==0  (gaussian_diffusion.py:696-715)."""
    tb = torch.full((x.shape[0],), t, dtype=torch.long)
    x0 = predict_start_from_noise(buf, x, tb, eps_model(x, tb)).clamp(-1.0, 1.0)
    mean, _, logvar = q_posterior(buf, x0, x, tb)
    if t > 0:
        return mean + (0.5 * logvar).exp() * noise, x0
    return mean, x0


def p_sample_loop(buf, eps_model, x_T: Tensor, noises: Sequence[Tensor]) -> Tensor:
    """T sequential p_sample steps; noises[i] is the i-th randn_like drawn (t = T-1 .. 1)  (gaussian_diffusion.py:717-733)."""
    T = buf["betas"].shape[0]
    img = x_T
    for i, t in enumerate(reversed(range(T))):
        img, _ = p_sample(buf, eps_model, img, t, noises[i] if t > 0 else None)
    return img


def ddim_sample(buf, eps_model, x_T: Tensor, S: int, eta: float, noises: Sequence[Tensor]) -> Tensor:
    """DDIM with clip_x_start=True  (gaussian_diffusion.py:735-777)."""
    T = buf["betas"].shape[0]
    times = list(reversed(torch.linspace(-1, T - 1, steps=S + 1).int().tolist()))
    img = x_T
    ac = buf["alphas_cumprod"]
    j = 0
    for time, time_next in zip(times[:-1], times[1:]):
        tb = torch.full((img.shape[0],), time, dtype=torch.long)
        pred = eps_model(img, tb)
        x0 = predict_start_from_noise(buf, img, tb, pred).clamp(-1.0, 1.0)
        if time_next < 0:
            img = x0
            continue
        a, an = ac[time], ac[time_next]
        sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
        c = (1 - an - sigma ** 2).sqrt()
        img = x0 * an.sqrt() + c * pred + sigma * noises[j]
        j += 1
    return img


def p_losses(buf, eps_model, x0: Tensor, t: Tensor, noise: Tensor, loss_type: str = "l2") -> Tensor:
    """mean_b( mean_chw(|eps_hat - eps|^p) * p2w[t] ), objective pred_noise  (gaussian_diffusion.py:823-859)."""
    x = q_sample(buf, x0, t, noise)
    out = eps_model(x, t)
    if loss_type == "l2":
        l = (out - noise) ** 2
    elif loss_type == "l1":
        l = (out - noise).abs()
    else:
        raise ValueError(f"invalid loss type {loss_type}")
    l = l.reshape(l.shape[0], -1).mean(dim=1) * buf["p2_loss_weight"].gather(-1, t)
    return l.mean()


# ----------------------------------------------------------------------------- variational bound (NLL evaluation)
def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL between diagonal normals, reference operation order  (gaussian_diffusion.py:36-44)."""
    if not isinstance(logvar2, torch.Tensor):
        logvar2 = torch.tensor(logvar2).repeat(mean1.shape[0]).reshape(-1, 1, 1, 1)
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * torch.exp(-logvar2))


def gaussian_log_likelihood(x: Tensor, means: Tensor, log_scales: Tensor) -> Tensor:
    """log of the N(means, e^{log_scales}) mass of [x-0.1, x+0.1] through the tanh CDF approximation, clamped at 1e-12
    (gaussian_diffusion.py:47-74)."""
    def cdf(v):
        return 0.5 * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (v + 0.044715 * torch.pow(v, 3))))
    centered = x - means
    inv_stdv = torch.exp(-log_scales)
    delta = cdf(inv_stdv * (centered + 1e-1)) - cdf(inv_stdv * (centered - 1e-1))
    return torch.log(delta.clamp(min=1e-12))


def _mean_flat(v: Tensor) -> Tensor:
    return v.mean(dim=list(range(1, v.ndim)))


def prior_bpd(buf, x0: Tensor) -> Tensor:
    """KL(q(x_T|x_0) || N(0,I)) per sample  (gaussian_diffusion.py:905-930)."""
    T = buf["betas"].shape[0]
    t = torch.full((x0.shape[0],), T - 1, dtype=torch.long)
    mean = _ext(buf["sqrt_alphas_cumprod"], t, x0.ndim) * x0
    logvar = _ext(buf["log_one_minus_alphas_cumprod"], t, x0.ndim)
    return _mean_flat(normal_kl(mean, logvar, 0.0, 0.0))


def vb_term(buf, eps_model, x0: Tensor, x_t: Tensor, t: Tensor, clip_denoised: bool = True) -> Tensor:
    """One term of the bound: decoder NLL at t == 0, posterior KL otherwise  (gaussian_diffusion.py:932-958)."""
    true_mean, _, true_logvar = q_posterior(buf, x0, x_t, t)
    x0_hat = predict_start_from_noise(buf, x_t, t, eps_model(x_t, t))
    if clip_denoised:
        x0_hat = x0_hat.clamp(-1.0, 1.0)
    model_mean, _, model_logvar = q_posterior(buf, x0_hat, x_t, t)
    kl = _mean_flat(normal_kl(true_mean, true_logvar, model_mean, model_logvar))
    nll = _mean_flat(-gaussian_log_likelihood(x0, model_mean, 0.5 * model_logvar))
    return torch.where(t == 0, nll, kl)


def calc_neg_log_likelihood_loop(buf, eps_model, x0: Tensor, noises: Sequence[Tensor]) -> Tensor:
    """sum_t vb_term(t) + prior term, t = T-1 .. 0; noises[i] is the i-th randn_like drawn  (gaussian_diffusion.py:869-903)."""
    T = buf["betas"].shape[0]
    vb = []
    for i, t in enumerate(reversed(range(T))):
        tb = torch.full((x0.shape[0],), t, dtype=torch.long)
        vb.append(vb_term(buf, eps_model, x0, q_sample(buf, x0, tb, noises[i]), tb))
    return torch.stack(vb, dim=1).sum(dim=1) + prior_bpd(buf, x0)


# ----------------------------------------------------------------------------- objectives / self-conditioning (variants)
def predict_noise_from_start(buf, x_t: Tensor, t: Tensor, x0: Tensor) -> Tensor:
    """(gaussian_diffusion.py:647-651)."""
    return (_ext(buf["sqrt_recip_alphas_cumprod"], t, x_t.ndim) * x_t - x0) / _ext(buf["sqrt_recipm1_alphas_cumprod"], t, x_t.ndim)


def predict_v(buf, x_start: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """(gaussian_diffusion.py:653-657)."""
    return _ext(buf["sqrt_alphas_cumprod"], t, x_start.ndim) * noise - _ext(buf["sqrt_one_minus_alphas_cumprod"], t, x_start.ndim) * x_start


def predict_start_from_v(buf, x_t: Tensor, t: Tensor, v: Tensor) -> Tensor:
    """(gaussian_diffusion.py:659-663)."""
    return _ext(buf["sqrt_alphas_cumprod"], t, x_t.ndim) * x_t - _ext(buf["sqrt_one_minus_alphas_cumprod"], t, x_t.ndim) * v


def model_predictions(buf, model, x: Tensor, t: Tensor, objective: str = "pred_noise", x_self_cond: Optional[Tensor] = None,
                      clip_x_start: bool = False):
    """-> (pred_noise, x_start) for the three objectives  (gaussian_diffusion.py:674-694).  model(x, t, x_self_cond)."""
    out = model(x, t, x_self_cond)
    clip = (lambda v: v.clamp(-1.0, 1.0)) if clip_x_start else (lambda v: v)
    if objective == "pred_noise":
        return out, clip(predict_start_from_noise(buf, x, t, out))
    if objective == "pred_x0":
        x0 = clip(out)
    elif objective == "pred_v":
        x0 = clip(predict_start_from_v(buf, x, t, out))
    else:
        raise ValueError(f"unknown objective {objective}")
    return predict_noise_from_start(buf, x, t, x0), x0


def p_sample_v(buf, model, x: Tensor, t: int, noise: Optional[Tensor], objective: str = "pred_noise",
               x_self_cond: Optional[Tensor] = None):
    """p_sample for any objective / with self-conditioning  (gaussian_diffusion.py:696-715)."""
    tb = torch.full((x.shape[0],), t, dtype=torch.long)
    _, x0 = model_predictions(buf, model, x, tb, objective, x_self_cond)
    x0 = x0.clamp(-1.0, 1.0)
    mean, _, logvar = q_posterior(buf, x0, x, tb)
    return (mean + (0.5 * logvar).exp() * noise if t > 0 else mean), x0


def p_sample_loop_v(buf, model, x_T: Tensor, noises: Sequence[Tensor], objective: str = "pred_noise", self_condition: bool = False):
    """(gaussian_diffusion.py:717-733): with self-conditioning every step is conditioned on the previous step's x_start."""
    T = buf["betas"].shape[0]
    img, x_start = x_T, None
    for i, t in enumerate(reversed(range(T))):
        img, x_start = p_sample_v(buf, model, img, t, noises[i] if t > 0 else None, objective, x_start if self_condition else None)
    return img


def ddim_sample_v(buf, model, x_T: Tensor, S: int, eta: float, noises: Sequence[Tensor], objective: str = "pred_noise",
                  self_condition: bool = False) -> Tensor:
    """(gaussian_diffusion.py:735-777) for any objective / with self-conditioning."""
    T = buf["betas"].shape[0]
    times = list(reversed(torch.linspace(-1, T - 1, steps=S + 1).int().tolist()))
    img, x_start, ac, j = x_T, None, buf["alphas_cumprod"], 0
    for time, time_next in zip(times[:-1], times[1:]):
        tb = torch.full((img.shape[0],), time, dtype=torch.long)
        pred_noise, x_start = model_predictions(buf, model, img, tb, objective, x_start if self_condition else None, clip_x_start=True)
        if time_next < 0:
            img = x_start
            continue
        a, an = ac[time], ac[time_next]
        sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
        c = (1 - an - sigma ** 2).sqrt()
        img = x_start * an.sqrt() + c * pred_noise + sigma * noises[j]
        j += 1
    return img


def p_losses_v(buf, model, x0: Tensor, t: Tensor, noise: Tensor, objective: str = "pred_noise", loss_type: str = "l2",
               self_condition: bool = False, random_draw: float = 1.0) -> Tensor:
    """(gaussian_diffusion.py:823-859): target by objective; self-conditioning when random_draw < 0.5 (the reference calls random())."""
    x = q_sample(buf, x0, t, noise)
    x_self_cond = None
    if self_condition and random_draw < 0.5:
        with torch.no_grad():
            x_self_cond = model_predictions(buf, model, x, t, objective)[1].detach()
    out = model(x, t, x_self_cond)
    target = {"pred_noise": noise, "pred_x0": x0}.get(objective)
    if target is None:
        target = predict_v(buf, x0, t, noise)
    l = (out - target) ** 2 if loss_type == "l2" else (out - target).abs()
    l = l.reshape(l.shape[0], -1).mean(dim=1) * buf["p2_loss_weight"].gather(-1, t)
    return l.mean()
