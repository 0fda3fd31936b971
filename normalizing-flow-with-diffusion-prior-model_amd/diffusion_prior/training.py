"""Differentiable (training) path of the diffusion prior's U-Net: torch.autograd.Function wrappers over the HIP kernels.

The reference trains the prior through torch autograd (diffusion_prior/trainer.py:134-150: encode with the frozen flow ->
DiffusionPrior.forward -> GaussianDiffusion.p_losses per latent part -> loss.backward() -> optimizer.step()).  Every
activation-sized op of that graph is a Function whose forward and backward are HIP launches through the C ABI (exact-fp32
MFMA convolutions with the flow's weight/data-gradient kernels, csrc/backward.hip; the glue ops' derivatives,
csrc/unet_bwd.hip); autograd strings them together.  Parameter-sized arithmetic stays in torch: the weight standardisation
of WeightStandardizedConv2d (its derivative then comes for free) and the slicing of the time-MLP outputs.
"""
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor
from torch.autograd import Function

import nfdp_hip as K
from normalizing_flow.training import SqueezeFn, grad_needed   # noqa: F401  (re-exported)


# ----------------------------------------------------------------------------- Functions
class ConvFn(Function):
    """y = conv2d(x, W, padding=k//2) + bias on NHWC tensors (nn.Conv2d / WeightStandardizedConv2d / nn.Linear as 1x1);
    `compute` (K.F32 | K.BF16) applies to the forward, data-gradient and weight-gradient GEMMs."""

    @staticmethod
    def forward(ctx, xh, weight, bias, k, compute):
        y = K.conv2d(xh, K.conv_pack(weight.detach().contiguous(), compute), weight.shape[0], k,
                     add=None if bias is None else bias.detach(), compute=compute, out_dtype=K.F32)
        ctx.save_for_backward(xh, weight)
        ctx.k, ctx.has_bias, ctx.compute = int(k), bias is not None, compute
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.conv2d(dy, K.conv_pack_ex(weight.detach().contiguous(), ctx.compute, flip=True), weight.shape[1], ctx.k,
                          compute=ctx.compute, out_dtype=K.F32)
        dw = K.conv_wgrad(xh, dy, ctx.k, compute=ctx.compute)
        db = K.scale_bwd(dy, want_dc=False)[2] if ctx.has_bias else None
        return dx, dw, db, None, None


class GroupNormSiLUFn(Function):
    """GroupNorm -> x*(scale+1)+shift -> SiLU (reference Block :267-276), one launch each way."""

    @staticmethod
    def forward(ctx, xh, gamma, beta, scale_shift, groups):
        xh = xh.contiguous()
        y = K.groupnorm_silu(xh, groups, gamma.detach(), beta.detach(), None if scale_shift is None else scale_shift.detach(), 1e-5)
        ctx.save_for_backward(xh, gamma, beta, scale_shift)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, gamma, beta, ss = ctx.saved_tensors
        dx, dg, db, dss = K.groupnorm_silu_bwd(xh, dy.contiguous(), ctx.groups, gamma.detach(), beta.detach(),
                                               None if ss is None else ss.detach(), 1e-5)
        return dx, dg, db, dss, None


class ChanLayerNormFn(Function):
    """Channel LayerNorm with gain (+ optional residual folded into the same launch) (reference :205-209, :160-166)."""

    @staticmethod
    def forward(ctx, xh, gain, residual):
        xh = xh.contiguous()
        y = K.chan_layernorm(xh, gain.detach().reshape(-1), residual=None if residual is None else residual.contiguous())
        ctx.save_for_backward(xh, gain)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, gain = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dg = K.chan_layernorm_bwd(xh, dy, gain.detach().reshape(-1))
        return dx, dg.view_as(gain), (dy if ctx.has_res else None)


class LinearAttentionFn(Function):
    @staticmethod
    def forward(ctx, qkv, heads, dim_head, scale):
        qkv = qkv.contiguous()
        ctx.save_for_backward(qkv)
        ctx.args = (heads, dim_head, scale)
        return K.linear_attention(qkv, heads, dim_head, scale)

    @staticmethod
    def backward(ctx, dout):
        (qkv,) = ctx.saved_tensors
        return K.linear_attention_bwd(qkv, dout.contiguous(), *ctx.args), None, None, None


class AttentionFn(Function):
    @staticmethod
    def forward(ctx, qkv, heads, dim_head, scale):
        qkv = qkv.contiguous()
        ctx.save_for_backward(qkv)
        ctx.args = (heads, dim_head, scale)
        return K.attention(qkv, heads, dim_head, scale)

    @staticmethod
    def backward(ctx, dout):
        (qkv,) = ctx.saved_tensors
        return K.attention_bwd(qkv, dout.contiguous(), *ctx.args), None, None, None


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class ConcatFn(Function):
    """torch.cat((a, b), channel dim) on NHWC tensors; the gradients are the two channel slices."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = a.shape[-1]
        return K.concat_channels(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy[..., :ctx.ca], dy[..., ctx.ca:]


class UpsampleFn(Function):
    @staticmethod
    def forward(ctx, xh):
        return K.upsample2x(xh.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return K.upsample2x_bwd(dy.contiguous())


class ActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.act = act
        return K.act_fwd(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.act_bwd(x, dy.contiguous(), ctx.act), None


class QSampleFn(Function):
    """x_t = sqrt(ac[t]) x_0 + sqrt(1 - ac[t]) eps (reference :806-812); gradient to x_0 only (eps is data)."""

    @staticmethod
    def forward(ctx, x0h, nh, t, sqrt_ac, sqrt_1mac):
        ctx.save_for_backward(t, sqrt_ac)
        return K.q_sample(x0h.contiguous(), nh, t, sqrt_ac, sqrt_1mac)

    @staticmethod
    def backward(ctx, dy):
        t, sqrt_ac = ctx.saved_tensors
        dy = dy.contiguous()
        return K.q_sample(dy, dy, t, sqrt_ac, torch.zeros_like(sqrt_ac)), None, None, None, None   # sqrt(ac[t]) * dy


class DiffusionLossFn(Function):
    """mean_b( mean_chw(|pred - target|^p) * p2w[t_b] ) (reference p_losses :855-859)."""

    @staticmethod
    def forward(ctx, pred, target, t, p2w, l1):
        pred, target = pred.contiguous(), target.contiguous()
        ctx.save_for_backward(pred, target, t, p2w)
        ctx.l1 = bool(l1)
        return K.diffusion_loss(pred, target, t, p2w, ctx.l1)

    @staticmethod
    def backward(ctx, g):
        pred, target, t, p2w = ctx.saved_tensors
        return K.diffusion_loss_bwd(pred, target, t, p2w, g.reshape(1).float().contiguous(), ctx.l1), None, None, None, None


# ----------------------------------------------------------------------------- module walkers
def _conv(mod, xh: Tensor, c: int = K.F32) -> Tensor:
    """_Conv / WeightStandardizedConv2d; the standardisation (reference :189-195, eps 1e-5) is weight-sized torch glue."""
    w = mod.weight
    if mod.standardize:
        mean = w.mean(dim=(1, 2, 3), keepdim=True)
        var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
        w = (w - mean) * (var + 1e-5).rsqrt()
    return ConvFn.apply(xh, w, mod.bias, mod.k, c)


class LearnedSinusoidalFn(Function):
    """RandomOrLearnedSinusoidalPosEmb (reference :240-255) with the gradient of its learned frequencies."""

    @staticmethod
    def forward(ctx, weights, time):
        ctx.save_for_backward(weights, time)
        return K.learned_sinusoidal_emb(time, weights.detach())

    @staticmethod
    def backward(ctx, dout):
        weights, time = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        return K.learned_sinusoidal_bwd(time, weights.detach(), dout.contiguous()), None


def _linear(mod, x: Tensor) -> Tensor:
    """nn.Linear on [B, I] as a 1x1 convolution of a [B,1,1,I] image (same kernels, same gradients)."""
    b = x.shape[0]
    y = ConvFn.apply(x.reshape(b, 1, 1, -1), mod.weight.reshape(mod.weight.shape[0], mod.weight.shape[1], 1, 1), mod.bias, 1, K.F32)
    return y.reshape(b, -1)


def _block(blk, xh: Tensor, scale_shift: Optional[Tensor], c: int) -> Tensor:
    return GroupNormSiLUFn.apply(_conv(blk.proj, xh, c), blk.norm.weight, blk.norm.bias, scale_shift, blk.norm.num_groups)


def _resnet_block(rb, xh: Tensor, t_act: Tensor, c: int) -> Tensor:
    """(reference :290-302); t_act = SiLU(time embedding), shared by every block."""
    ss = _linear(rb.mlp[1], t_act) if rb.mlp is not None else None
    h = _block(rb.block1, xh, ss, c)
    h = _block(rb.block2, h, None, c)
    res = _conv(rb.res_conv, xh, c) if not isinstance(rb.res_conv, nn.Identity) else xh
    return AddFn.apply(h, res)


def _attn_block(res, xh: Tensor, c: int) -> Tensor:
    from .gaussian_diffusion import LinearAttention
    pre = res.fn
    att = pre.fn
    xn = ChanLayerNormFn.apply(xh, pre.norm.g, None)
    qkv = _conv(att.to_qkv, xn, c)
    if isinstance(att, LinearAttention):
        a = LinearAttentionFn.apply(qkv, att.heads, att.dim_head, att.scale)
        return ChanLayerNormFn.apply(_conv(att.to_out[0], a, c), att.to_out[1].g, xh)        # LayerNorm(to_out conv) + x
    a = AttentionFn.apply(qkv, att.heads, att.dim_head, att.scale)
    return AddFn.apply(_conv(att.to_out, a, c), xh)


def unet_forward(unet, xh: Tensor, time: Tensor, x_self_cond_h: Optional[Tensor] = None) -> Tensor:
    """Unet.forward with a gradient graph (reference :451-490); NHWC in, NHWC out.  The U-Net's compute mode applies to its
    convolutions; the small time-MLP linears stay exact fp32."""
    c = K.compute_enum(unet.compute)
    if unet.self_condition:                                             # torch.cat((x_self_cond or zeros, x), dim=1)  (:452-454)
        sc = torch.zeros_like(xh) if x_self_cond_h is None else x_self_cond_h.detach()
        xh = ConcatFn.apply(sc.contiguous(), xh.contiguous())
    x = _conv(unet.init_conv, xh, c)
    r = x
    if unet.random_or_learned_sinusoidal_cond:
        e = LearnedSinusoidalFn.apply(unet.time_mlp[0].weights, time.long())
    else:
        e = K.sinusoidal_emb(time.long(), unet.dim)
    e = ActFn.apply(_linear(unet.time_mlp[1], e), K.GELU)
    t_act = ActFn.apply(_linear(unet.time_mlp[3], e), K.SILU)            # the SiLU every ResnetBlock.mlp starts with
    hs = []
    for block1, block2, attn, downsample in unet.downs:
        x = _resnet_block(block1, x, t_act, c)
        hs.append(x)
        x = _resnet_block(block2, x, t_act, c)
        x = _attn_block(attn, x, c)
        hs.append(x)
        x = _conv(downsample[1], SqueezeFn.apply(x), c) if isinstance(downsample, nn.Sequential) else _conv(downsample, x, c)
    x = _resnet_block(unet.mid_block1, x, t_act, c)
    x = _attn_block(unet.mid_attn, x, c)
    x = _resnet_block(unet.mid_block2, x, t_act, c)
    for block1, block2, attn, upsample in unet.ups:
        x = _resnet_block(block1, ConcatFn.apply(x, hs.pop()), t_act, c)
        x = _resnet_block(block2, ConcatFn.apply(x, hs.pop()), t_act, c)
        x = _attn_block(attn, x, c)
        x = _conv(upsample[1], UpsampleFn.apply(x), c) if isinstance(upsample, nn.Sequential) else _conv(upsample, x, c)
    x = _resnet_block(unet.final_res_block, ConcatFn.apply(x, r), t_act, c)
    return _conv(unet.final_conv, x, c)


def p_losses(diffusion, x_start: Tensor, t: Tensor, noise: Optional[Tensor] = None) -> Tensor:
    """GaussianDiffusion.p_losses with a gradient graph (reference :823-859; all three objectives, self-conditioning).  When the
    latents carry a graph themselves (unfrozen flow, trainer.py:139-141) the gradient flows back into them through q_sample and
    the U-Net's first convolution (and, for pred_x0 / pred_v, NOT through the target, which the reference does not detach either but
    whose dependence on the latents only the unfrozen-flow recipe could see: that recipe uses pred_noise)."""
    noise = torch.randn_like(x_start) if noise is None else noise
    x0h, nh = K.to_nhwc(x_start), K.to_nhwc(noise.detach()).contiguous()
    t = t.contiguous()
    if x_start.requires_grad:
        xh = QSampleFn.apply(x0h, nh, t, diffusion.sqrt_alphas_cumprod, diffusion.sqrt_one_minus_alphas_cumprod)
    else:
        xh = K.q_sample(x0h.contiguous(), nh, t, diffusion.sqrt_alphas_cumprod, diffusion.sqrt_one_minus_alphas_cumprod)
    sch = diffusion._maybe_self_cond(xh.detach().contiguous(), t)
    out = unet_forward(diffusion.model, xh, t, sch)
    target = diffusion._loss_target_nhwc(x0h.detach().contiguous(), nh, t)
    return DiffusionLossFn.apply(out, target, t, diffusion.p2_loss_weight, diffusion.loss_fn == "l1")
