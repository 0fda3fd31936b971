"""U-Net eps-predictor and Gaussian diffusion on the MI355X kernel library — mirrors the hot-path symbols of the
reference's diffusion_prior/gaussian_diffusion.py (a trimmed denoising-diffusion-pytorch):
  Unet :363-490 (+ blocks :160-358), schedules :501-537, GaussianDiffusion :540-867.

Same constructor arguments, attribute names, method names/returns and the same module tree, so state_dict keys are
identical.  Modules hold parameters in the reference's layouts; `forward` runs hand-written HIP kernels on NHWC
tensors through the C ABI (convs: MFMA implicit GEMM with weight standardisation folded into the weight pack;
GroupNorm+scale/shift+SiLU, channel LayerNorm(+residual), linear/full attention cores, time MLP, and the fused
reverse-diffusion step with on-device schedule lookup are one launch each).  All randomness can be injected
(`noise=` / `noises=`) for reproducible parity tests.  With trainable parameters and grad mode on, `Unet.forward` and
`p_losses` run the differentiable path of diffusion_prior/training.py (autograd Functions over HIP forward/backward kernels).

All three objectives ('pred_noise', 'pred_x0', 'pred_v'), self-conditioning and auto_normalize are built (the captured-graph loops
serve the plain case - every recipe of the reference - and the eager loops the self-conditioned one).  The U-Net also takes the
learned / random sinusoidal time embeddings (:240-255); GaussianDiffusion rejects such a U-Net exactly as the reference does (:559).
"""
import math
from random import random as _random
from collections import namedtuple
from functools import partial
from typing import Optional, Sequence

import numpy as np
import torch
from torch import nn, Tensor

import nfdp_hip as K
from normalizing_flow.base import _EpochOnApply

ModelPrediction = namedtuple('ModelPrediction', ['pred_noise', 'pred_x_start'])


# The eight small helpers below (exists ... unnormalize_to_zero_to_one's formulas), `extract`, the three beta schedules and the
# schedule dispatch in GaussianDiffusion.__init__ follow the reference line for line (gaussian_diffusion.py:110-124, :148-157,
# :495-539, :573-580, itself a trimmed copy of lucidrains' denoising-diffusion-pytorch 0.31, MIT): one-off host code and constant
# tables behind a class surface the drivers fix.  Everything that touches the device is written around the HIP calls.
def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def identity(t, *args, **kwargs):
    return t


def normalize_to_neg_one_to_one(img):
    """img * 2 - 1  (reference :148-150).  An input that carries a gradient (latents of an unfrozen flow, trainer.py:139-141) keeps it:
    the HIP launch sits outside autograd, so that case takes the same two fp32 operations as torch ops."""
    if torch.is_grad_enabled() and img.requires_grad:
        return img * 2.0 - 1.0
    return K.as_nchw(K.axpb(K.to_nhwc(img), 2.0, -1.0))


def unnormalize_to_zero_to_one(t):
    """(t + 1) * 0.5  (reference :153-157)"""
    if torch.is_grad_enabled() and t.requires_grad:
        return (t + 1.0) * 0.5
    return K.as_nchw(K.axpb(K.to_nhwc(t), 0.5, 1.0, add_first=True))


# ----------------------------------------------------------------------------- parameter holders
class _Conv(nn.Module):
    """nn.Conv2d's parameter names/shapes/default init; forward = MFMA implicit GEMM (+ optional weight standardisation)."""

    def __init__(self, cin, cout, k, bias=True, standardize=False):
        super().__init__()
        self.cin, self.cout, self.k, self.standardize = cin, cout, k, standardize
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        bound = 1.0 / math.sqrt(cin * k * k)
        nn.init.uniform_(self.weight, -bound, bound)
        if bias:
            self.bias = nn.Parameter(torch.empty(cout))
            nn.init.uniform_(self.bias, -bound, bound)
        else:
            self.register_parameter("bias", None)
        self._key, self._packed = None, None

    def packed(self, compute: int) -> Tensor:
        key = (self.weight.data_ptr(), self.weight._version, compute, K.param_epoch())
        if key != self._key:
            w = self.weight.detach()
            if self.standardize:                       # WeightStandardizedConv2d.forward :189-195, eps 1e-5 (fp32)
                w = K.ws_standardize(w, 1e-5)
            self._packed, self._key = K.conv_pack(w, compute), key
        return self._packed

    def packed_k7(self) -> Tensor:
        """Unet.init_conv's weight in the layout of the direct 7x7 kernel (K.conv_k7_pack)."""
        key = (self.weight.data_ptr(), self.weight._version, K.param_epoch())
        if key != getattr(self, "_key_k7", None):
            self._packed_k7, self._key_k7 = K.conv_k7_pack(self.weight.detach()), key
        return self._packed_k7

    def packed2x2(self) -> Tensor:
        """Downsample's 1x1 weight [Cout][4C] as the 2x2 / stride-2 weight [Cout][C][2][2] it is on the un-rearranged tensor
        (rearranged channel = c * 4 + p1 * 2 + p2, reference :176-180), packed fp32."""
        key = (self.weight.data_ptr(), self.weight._version, K.param_epoch())
        if key != getattr(self, "_key22", None):
            self._packed22, self._key22 = K.conv_pack(self.weight.detach().view(self.cout, self.cin // 4, 2, 2), K.F32), key
        return self._packed22

    def forward_nhwc(self, xh: Tensor, compute: Optional[str] = None, x2: Optional[Tensor] = None) -> Tensor:
        """x2: the input is torch.cat((xh, x2), dim=channels), read from the two tensors (exact-fp32 inference; K.cat_sources_ok)."""
        c = K.compute_enum(compute)
        if x2 is not None:
            return K.conv2d_cat(xh, x2, self.packed(c), self.cout, self.k, add=self.bias)
        return K.conv2d(xh, self.packed(c), self.cout, self.k, add=self.bias, compute=c, out_dtype=K.F32)


class WeightStandardizedConv2d(_Conv):
    """(reference :183-197)"""

    def __init__(self, cin, cout, k, padding=1):
        super().__init__(cin, cout, k, bias=True, standardize=True)


class _Linear(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        bound = 1.0 / math.sqrt(fin)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


class _GroupNormParams(nn.Module):
    def __init__(self, groups, dim):
        super().__init__()
        self.num_groups = groups
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Marker(nn.Module):
    """Parameter-free placeholder keeping the reference's nn.Sequential indices (Rearrange / nn.Upsample / SiLU / GELU)."""

    def __init__(self, what):
        super().__init__()
        self.what = what


# ----------------------------------------------------------------------------- small helper modules (:160-255)
class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class LayerNorm(nn.Module):
    """Channel LayerNorm with a gain only  (reference :200-209)."""

    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = LayerNorm(dim)


def Upsample(dim, dim_out=None):
    return nn.Sequential(_Marker("nearest2x"), _Conv(dim, default(dim_out, dim), 3))


def Downsample(dim, dim_out=None):
    return nn.Sequential(_Marker("squeeze2x2"), _Conv(dim * 4, default(dim_out, dim), 1))


class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim


class RandomOrLearnedSinusoidalPosEmb(nn.Module):
    """[t | sin(2 pi t w) | cos(2 pi t w)] with random (frozen) or learned frequencies w  (reference :240-255; same parameter name)."""

    def __init__(self, dim, is_random=False):
        super().__init__()
        assert (dim % 2) == 0
        self.weights = nn.Parameter(torch.randn(dim // 2), requires_grad=not is_random)

    def forward(self, x: Tensor) -> Tensor:
        return K.learned_sinusoidal_emb(x.long(), self.weights.detach())


class Block(nn.Module):
    """WS-conv3x3 -> GroupNorm -> x*(scale+1)+shift -> SiLU  (reference :260-276)."""

    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.proj = WeightStandardizedConv2d(dim, dim_out, 3, padding=1)
        self.norm = _GroupNormParams(groups, dim_out)
        self.act = _Marker("silu")

    def forward_nhwc(self, xh, scale_shift=None, compute=None, residual=None, x2=None):
        cenum = K.compute_enum(compute)
        if cenum == K.F32 and not torch.is_grad_enabled() and \
                K.conv2d_gn_silu_supported(xh, self.proj.cout, self.proj.k, self.norm.num_groups, self.proj.bias, self.norm.weight, self.norm.bias,
                                           extra_cin=0 if x2 is None else x2.shape[3]):
            # the whole Block as ONE launch: the conv's summed tile is normalised in its epilogue (8x8 / 4x4 / 2x2 stages)
            return K.conv2d_gn_silu(xh, self.proj.packed(cenum), self.proj.cout, self.proj.k, self.proj.bias, self.norm.num_groups,
                                    self.norm.weight, self.norm.bias, scale_shift, 1e-5, residual, x2=x2)
        h = self.proj.forward_nhwc(xh, compute, x2=x2)
        return K.groupnorm_silu(h, self.norm.num_groups, self.norm.weight, self.norm.bias, scale_shift, 1e-5, residual)


class ResnetBlock(nn.Module):
    """(reference :279-302)"""

    def __init__(self, dim, dim_out, *, time_emb_dim=None, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(_Marker("silu"), _Linear(time_emb_dim, dim_out * 2)) if exists(time_emb_dim) else None
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.res_conv = _Conv(dim, dim_out, 1) if dim != dim_out else nn.Identity()

    def forward_nhwc(self, xh, temb=None, compute=None, scale_shift=None):
        """scale_shift: this block's precomputed [B, scale | shift] rows (the U-Net batches every block's time-MLP
        into one GEMM); otherwise it is computed here from temb (SiLU -> Linear, reference :290-295)."""
        ss = scale_shift
        if ss is None and exists(self.mlp) and exists(temb):
            ss = K.linear(temb, self.mlp[1].weight, self.mlp[1].bias, pre_act=1)
        x2 = None
        if isinstance(xh, tuple):
            # (x, skip): the reference's torch.cat((x, h.pop()), dim=1) (:479-486).  In exact-fp32 inference the two convs that read it
            # (block1's 3x3 and the 1x1 res_conv: dim != dim_out here) read the two tensors; otherwise it is materialised.
            xh, x2 = xh
            if not (K.compute_enum(compute) == K.F32 and not torch.is_grad_enabled() and isinstance(self.res_conv, _Conv)
                    and K.cat_sources_ok(xh, x2)):
                xh, x2 = K.concat_channels(xh, x2), None
        b1, b2 = self.block1, self.block2
        if K.compute_enum(compute) == K.F32 and not torch.is_grad_enabled() and b2.proj.k == 3:
            # where a conv workgroup's tile is a part of an image (the one-launch Block does not apply) block1's GroupNorm rides in
            # block2's conv: block1's conv leaves the group sums, block2's conv normalises while it stages its input tile
            e2 = 0 if x2 is None else x2.shape[3]
            p1 = (b1.proj.bias, b1.norm.weight, b1.norm.bias)
            rc0 = self.res_conv
            if isinstance(rc0, _Conv) and b1.proj.k == 3 and \
                    K.conv2d_gn_silu_res_supported(xh, b1.proj.cout, b1.norm.num_groups, *p1, rc0.bias, ss, extra_cin=e2):
                # tiles that hold whole images (4x4 / 2x2 stages): block1 as ONE launch, with the res_conv from a second accumulator set
                cenum = K.compute_enum(compute)
                h, res = K.conv2d_gn_silu_res(xh, b1.proj.packed(cenum), b1.proj.cout, b1.proj.bias, b1.norm.num_groups, b1.norm.weight,
                                              b1.norm.bias, ss, rc0.packed(cenum), rc0.bias, x2=x2)
                return b2.forward_nhwc(h, None, compute, residual=res)
            if not K.conv2d_gn_silu_supported(xh, b1.proj.cout, b1.proj.k, b1.norm.num_groups, *p1, extra_cin=e2):
                fused2 = K.conv2d_gn_silu_shape_supported(xh.shape[0], xh.shape[1], xh.shape[2], b2.proj.cin, b2.proj.cout, 3, b2.norm.num_groups)
                if K.conv2d_gn_carried_supported(xh, b1.proj.cout, b1.proj.k, b1.norm.num_groups, b2.norm.num_groups, fused2, *p1,
                                                 b2.proj.bias, b2.norm.weight, b2.norm.bias, ss, extra_cin=e2):
                    cenum = K.compute_enum(compute)
                    rc = self.res_conv
                    if isinstance(rc, _Conv) and b1.proj.k == 3 and K.conv2d_res_supported(xh, rc.cout, rc.bias, extra_cin=e2):
                        # ... and the 1x1 res_conv of the same input from a second accumulator set of the same launch
                        raw, sums, tile, res = K.conv2d_stats(xh, b1.proj.packed(cenum), b1.proj.cout, 3, b1.proj.bias, b1.norm.num_groups,
                                                              x2=x2, res=(rc.packed(cenum), rc.bias))
                    else:
                        raw, sums, tile = K.conv2d_stats(xh, b1.proj.packed(cenum), b1.proj.cout, b1.proj.k, b1.proj.bias, b1.norm.num_groups, x2=x2)
                        res = (rc.forward_nhwc(xh, compute, x2=x2) if isinstance(rc, _Conv) else xh).contiguous()
                    g1, g2 = b1.norm.num_groups, b2.norm.num_groups
                    nb, nh, nw = raw.shape[:3]
                    if (nb * nh * nw // 64) * -(-b2.proj.cout // 64) > K.GN_ON_LOAD_MAX_TILES and not fused2 and \
                            K.groupnorm_silu_sums_supported(nb, nh, nw, xh.shape[3] + e2, b1.proj.cout, g1, b1.norm.weight, b1.norm.bias, ss) and \
                            K.groupnorm_silu_sums_supported(nb, nh, nw, b2.proj.cin, b2.proj.cout, g2, b2.norm.weight, b2.norm.bias, res) and \
                            b1.proj.k == 3:
                        # more than one chip-wide wave of conv workgroups: normalising while staging repeats the norm's arithmetic per
                        # tile halo and output-channel tile (+47 us on a 62 us conv at 32 x 64 x 64 x 64) where the streaming norm that
                        # reads the conv's sums takes 13 us - both norms as launches of that kind, both convs leaving their sums
                        h1 = K.groupnorm_silu_sums(raw, sums, tile, g1, b1.norm.weight, b1.norm.bias, ss, 1e-5)
                        raw2, sums2, tile2 = K.conv2d_stats(h1, b2.proj.packed(cenum), b2.proj.cout, 3, b2.proj.bias, g2)
                        return K.groupnorm_silu_sums(raw2, sums2, tile2, g2, b2.norm.weight, b2.norm.bias, None, 1e-5, res)
                    if fused2:
                        return K.conv2d_gnin(raw, sums, tile, b1.norm.num_groups, b1.norm.weight, b1.norm.bias, ss, b2.proj.packed(cenum),
                                             b2.proj.cout, b2.proj.bias, out_groups=b2.norm.num_groups, out_gamma=b2.norm.weight,
                                             out_beta=b2.norm.bias, residual=res)
                    g2 = b2.norm.num_groups
                    if K.groupnorm_silu_sums_supported(raw.shape[0], raw.shape[1], raw.shape[2], b2.proj.cin, b2.proj.cout, g2, b2.norm.weight,
                                                       b2.norm.bias, res):
                        # block2's own norm: its conv leaves ITS group sums too, the norm is then a streaming pass without reductions
                        h2, sums2, tile2 = K.conv2d_gnin(raw, sums, tile, b1.norm.num_groups, b1.norm.weight, b1.norm.bias, ss,
                                                         b2.proj.packed(cenum), b2.proj.cout, b2.proj.bias, sums_groups=g2)
                        return K.groupnorm_silu_sums(h2, sums2, tile2, g2, b2.norm.weight, b2.norm.bias, None, 1e-5, res)
                    h2 = K.conv2d_gnin(raw, sums, tile, b1.norm.num_groups, b1.norm.weight, b1.norm.bias, ss, b2.proj.packed(cenum),
                                       b2.proj.cout, b2.proj.bias)
                    return K.groupnorm_silu(h2, g2, b2.norm.weight, b2.norm.bias, None, 1e-5, res)
        h = b1.forward_nhwc(xh, ss, compute, x2=x2)
        res = self.res_conv.forward_nhwc(xh, compute, x2=x2) if isinstance(self.res_conv, _Conv) else xh
        return b2.forward_nhwc(h, None, compute, residual=res.contiguous())   # h + res folded into block2's epilogue


class LinearAttention(nn.Module):
    """(reference :305-333)"""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        hidden_dim = dim_head * heads
        self.to_qkv = _Conv(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = nn.Sequential(_Conv(hidden_dim, dim, 1), LayerNorm(dim))


class Attention(nn.Module):
    """(reference :336-358)"""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        hidden_dim = dim_head * heads
        self.to_qkv = _Conv(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = _Conv(hidden_dim, dim, 1)


def _attn_block_nhwc(res: Residual, xh: Tensor, compute=None) -> Tensor:
    """Residual(PreNorm(dim, LinearAttention | Attention))  (reference :160-166, :212-220)."""
    pre = res.fn
    att = pre.fn
    c = K.compute_enum(compute)
    gpre = pre.norm.g.view(-1)
    hidden = att.heads * att.dim_head
    a = None
    lin = isinstance(att, LinearAttention)
    if c == K.F32 and K.linear_attention_qkv_supported(xh, att.heads, att.dim_head, gpre, linear=lin):
        # ... and to_qkv inside the attention core's launch: the qkv tensor is never written (two launches per block)
        a = K.linear_attention_qkv(xh, gpre, att.to_qkv.packed(c), att.heads, att.dim_head, att.scale, linear=lin)
    # no-grad exact-fp32 calls: the LayerNorms ride in the 1x1 convs' launches (nfdp_conv1x1_ln_f32: five launches per block -> three)
    elif c == K.F32 and K.conv1x1_ln_supported(xh, 3 * hidden, True, False, gpre):
        qkv = K.conv1x1_ln(xh, att.to_qkv.packed(c), 3 * hidden, pre_gain=gpre)
    else:
        qkv = att.to_qkv.forward_nhwc(K.chan_layernorm(xh, gpre), compute)
    if isinstance(att, LinearAttention):
        if a is None:
            a = K.linear_attention(qkv, att.heads, att.dim_head, att.scale)
        conv, gpost = att.to_out[0], att.to_out[1].g.view(-1)
        if c == K.F32 and K.conv1x1_ln_supported(a, conv.cout, False, True, conv.bias, gpost, xh):
            return K.conv1x1_ln(a, conv.packed(c), conv.cout, bias=conv.bias, post_gain=gpost, residual=xh.contiguous())
        o = conv.forward_nhwc(a, compute)
        return K.chan_layernorm(o, gpost, residual=xh)       # LayerNorm(to_out conv) + x
    if a is None:
        a = K.attention(qkv, att.heads, att.dim_head, att.scale)
    conv = att.to_out
    if c == K.F32 and K.conv1x1_ln_supported(a, conv.cout, False, False, conv.bias, xh):
        return K.conv1x1_ln(a, conv.packed(c), conv.cout, bias=conv.bias, residual=xh.contiguous())   # to_out + x in one launch
    return K.add(conv.forward_nhwc(a, compute), xh)


# ----------------------------------------------------------------------------- model (:363-490)
class Unet(_EpochOnApply, nn.Module):
    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16, compute: Optional[str] = None):
        super().__init__()
        self.compute = compute
        self.dim = dim
        self.channels = channels
        self.self_condition = self_condition
        input_channels = channels * (2 if self_condition else 1)
        init_dim = default(init_dim, dim)
        self.init_conv = _Conv(input_channels, init_dim, 7)
        dims = [init_dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))
        block_klass = partial(ResnetBlock, groups=resnet_block_groups)
        time_dim = dim * 4
        self.random_or_learned_sinusoidal_cond = learned_sinusoidal_cond or random_fourier_features
        if self.random_or_learned_sinusoidal_cond:                           # (reference :398-405)
            sinu_pos_emb, fourier_dim = RandomOrLearnedSinusoidalPosEmb(learned_sinusoidal_dim, random_fourier_features), learned_sinusoidal_dim + 1
        else:
            sinu_pos_emb, fourier_dim = SinusoidalPosEmb(dim), dim
        self.time_mlp = nn.Sequential(sinu_pos_emb, _Linear(fourier_dim, time_dim), _Marker("gelu"), _Linear(time_dim, time_dim))
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        num_resolutions = len(in_out)
        for ind, (dim_in, dim_out) in enumerate(in_out):
            is_last = ind >= (num_resolutions - 1)
            self.downs.append(nn.ModuleList([
                block_klass(dim_in, dim_in, time_emb_dim=time_dim),
                block_klass(dim_in, dim_in, time_emb_dim=time_dim),
                Residual(PreNorm(dim_in, LinearAttention(dim_in))),
                Downsample(dim_in, dim_out) if not is_last else _Conv(dim_in, dim_out, 3)
            ]))
        mid_dim = dims[-1]
        self.mid_block1 = block_klass(mid_dim, mid_dim, time_emb_dim=time_dim)
        self.mid_attn = Residual(PreNorm(mid_dim, Attention(mid_dim)))
        self.mid_block2 = block_klass(mid_dim, mid_dim, time_emb_dim=time_dim)
        for ind, (dim_in, dim_out) in enumerate(reversed(in_out)):
            is_last = ind == (len(in_out) - 1)
            self.ups.append(nn.ModuleList([
                block_klass(dim_out + dim_in, dim_out, time_emb_dim=time_dim),
                block_klass(dim_out + dim_in, dim_out, time_emb_dim=time_dim),
                Residual(PreNorm(dim_out, LinearAttention(dim_out))),
                Upsample(dim_out, dim_in) if not is_last else _Conv(dim_out, dim_in, 3)
            ]))
        default_out_dim = channels * (1 if not learned_variance else 2)
        self.out_dim = default(out_dim, default_out_dim)
        self.final_res_block = block_klass(dim * 2, dim, time_emb_dim=time_dim)
        self.final_conv = _Conv(dim, self.out_dim, 1)

    def time_embedding(self, time: Tensor, silu_out: bool = False) -> Tensor:
        """sinusoidal -> Linear -> GELU -> Linear  (reference :407-412); silu_out additionally applies the SiLU every
        ResnetBlock.mlp starts with (:282-283), once for all of them."""
        if self.random_or_learned_sinusoidal_cond:
            e = K.learned_sinusoidal_emb(time.long(), self.time_mlp[0].weights.detach())
        else:
            e = K.sinusoidal_emb(time.long(), self.dim)
        e = K.linear(e, self.time_mlp[1].weight, self.time_mlp[1].bias, post_act=1)
        return K.linear(e, self.time_mlp[3].weight, self.time_mlp[3].bias, post_act=2 if silu_out else 0)

    def _resnet_blocks(self):
        return [m for m in self.modules() if isinstance(m, ResnetBlock) and m.mlp is not None]

    def _all_scale_shifts(self, t_act: Tensor) -> dict:
        """Every ResnetBlock's time-MLP Linear (reference :284, 13 small GEMVs per forward) as ONE MFMA GEMM over the
        concatenated weights: [B,256] x [256, sum 2*dim_out] -> per-block column slices of one [B, sum] buffer."""
        blocks = self._resnet_blocks()
        c = K.compute_enum(self.compute)
        key = tuple((b.mlp[1].weight.data_ptr(), b.mlp[1].weight._version, b.mlp[1].bias._version) for b in blocks) + (c, K.param_epoch())
        if getattr(self, "_ss_key", None) != key:
            wcat = torch.cat([b.mlp[1].weight.detach() for b in blocks], dim=0)              # [sumO, 256] (plumbing)
            self._ss_bias = torch.cat([b.mlp[1].bias.detach() for b in blocks], dim=0).contiguous()
            self._ss_packed = K.conv_pack(wcat.reshape(wcat.shape[0], wcat.shape[1], 1, 1).contiguous(), c)
            self._ss_total = wcat.shape[0]
            self._ss_key = key
        bsz = t_act.shape[0]
        ss_all = K.conv2d(t_act.view(bsz, 1, 1, -1), self._ss_packed, self._ss_total, 1, add=self._ss_bias, compute=c,
                          out_dtype=K.F32).view(bsz, self._ss_total)
        return self._split_scale_shifts(ss_all)

    def _split_scale_shifts(self, ss_all: Tensor) -> dict:
        out, off = {}, 0
        for b in self._resnet_blocks():
            n = b.mlp[1].weight.shape[0]
            out[id(b)] = ss_all[:, off:off + n]
            off += n
        return out

    # ---- the time path as a table.  Everything between the integer timestep and the ResnetBlocks' (scale, shift) vectors - sinusoidal
    # embedding, two Linears, GELU, SiLU, the blocks' projections (reference :407-412, :282-284) - depends on t and the weights only, and a
    # diffusion process has T distinct timesteps.  Without gradients (sampling, the variational bound, inference losses) the four launches
    # per U-Net call (~38 us of a ~490 us reverse step at B = 64) become ONE row gather from a [T, sum 2*dim_out] table that is rebuilt
    # when a weight it depends on changes.
    def enable_time_table(self, num_timesteps: int) -> None:
        """Called by GaussianDiffusion: timesteps passed to this U-Net are integers in [0, num_timesteps)."""
        self._tt_T = max(int(num_timesteps), int(getattr(self, "_tt_T", 0) or 0))

    def _time_table(self, device) -> Optional[Tensor]:
        T_ = int(getattr(self, "_tt_T", 0) or 0)
        if T_ <= 0 or torch.is_grad_enabled() or device.type != "cuda":
            return None
        deps = getattr(self, "_tt_deps", None)
        if deps is None:                                  # the module structure is fixed after construction
            deps = self._tt_deps = list(self.time_mlp.parameters()) + [p for b in self._resnet_blocks() for p in b.mlp.parameters()]
        key = (T_, K.compute_enum(self.compute), K.param_epoch(), str(device)) + tuple((p.data_ptr(), p._version) for p in deps)
        if getattr(self, "_tt_key", None) != key:
            if torch.cuda.is_current_stream_capturing():
                return None                               # never build inside a capture (it would be rebuilt by every replay)
            t_all = torch.arange(T_, device=device, dtype=torch.long)
            self._tt = self._all_scale_shifts(self.time_embedding(t_all, silu_out=True))
            self._tt = torch.cat([self._tt[id(b)] for b in self._resnet_blocks()], dim=1).contiguous()
            self._tt_key = key
        return self._tt

    def _scale_shifts_for(self, time: Tensor) -> dict:
        table = self._time_table(time.device)
        if table is None:
            return self._all_scale_shifts(self.time_embedding(time, silu_out=True))
        return self._split_scale_shifts(table.index_select(0, time.long()))

    def with_self_cond(self, xh: Tensor, x_self_cond_h: Optional[Tensor]) -> Tensor:
        """torch.cat((x_self_cond or zeros, x), dim=1)  (reference :452-454), NHWC."""
        if not self.self_condition:
            return xh
        sc = torch.zeros_like(xh) if x_self_cond_h is None else x_self_cond_h
        return K.concat_channels(sc.contiguous(), xh.contiguous())

    def forward_nhwc(self, xh: Tensor, time: Tensor, x_self_cond_h: Optional[Tensor] = None) -> Tensor:
        c = self.compute
        xh = self.with_self_cond(xh, x_self_cond_h)
        ic = self.init_conv
        if K.compute_enum(c) == K.F32 and ic.k == 7 and K.conv_k7_narrow_supported(xh, ic.cout, ic.bias):
            x = K.conv_k7_narrow(xh, ic.packed_k7(), ic.cout, ic.bias)     # 7x7 over 1-4 latent channels: the direct fp32 kernel
        else:
            x = ic.forward_nhwc(xh, c)
        r = x
        ss = self._scale_shifts_for(time)
        rb = lambda blk, inp: blk.forward_nhwc(inp, None, c, scale_shift=ss[id(blk)])
        hs = []
        for block1, block2, attn, downsample in self.downs:
            x = rb(block1, x)
            hs.append(x)
            x = rb(block2, x)
            x = _attn_block_nhwc(attn, x, c)
            hs.append(x)
            if isinstance(downsample, nn.Sequential):
                dn = downsample[1]
                if K.compute_enum(c) == K.F32 and K.conv2d_squeeze_supported(x, dn.cout, dn.bias):
                    x = K.conv2d_squeeze(x, dn.packed2x2(), dn.cout, dn.bias)   # Rearrange(p1 p2) + 1x1 (:176-180) = a 2x2 stride-2 conv
                else:
                    x = dn.forward_nhwc(K.squeeze(x), c)
            else:
                x = downsample.forward_nhwc(x, c)
        x = rb(self.mid_block1, x)
        x = _attn_block_nhwc(self.mid_attn, x, c)
        x = rb(self.mid_block2, x)
        for block1, block2, attn, upsample in self.ups:
            x = rb(block1, (x, hs.pop()))                                # torch.cat((x, h.pop()), dim=1): read from both tensors
            x = rb(block2, (x, hs.pop()))
            x = _attn_block_nhwc(attn, x, c)
            if isinstance(upsample, nn.Sequential):
                up = upsample[1]
                if K.compute_enum(c) == K.F32 and K.conv2d_up2_supported(x, up.cout, up.bias):
                    x = K.conv2d_up2(x, up.packed(K.F32), up.cout, up.bias)   # nearest x2 + 3x3 (:169-173): the upsampling is an index map
                else:
                    x = up.forward_nhwc(K.upsample2x(x), c)
            else:
                x = upsample.forward_nhwc(x, c)
        x = rb(self.final_res_block, (x, r))
        return self.final_conv.forward_nhwc(x, c)

    def forward(self, x, time, x_self_cond=None):
        """(reference :451-490)"""
        sch = None if x_self_cond is None else K.to_nhwc(x_self_cond).contiguous()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from . import training                     # differentiable exact-fp32 path (diffusion_prior/training.py)
            return K.as_nchw(training.unet_forward(self, K.to_nhwc(x).contiguous(), time, sch))
        return K.as_nchw(self.forward_nhwc(K.to_nhwc(x).contiguous(), time, sch))


# ----------------------------------------------------------------------------- schedules (:495-537)
def extract(a, t, x_shape):
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


def linear_beta_schedule(timesteps):
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps, s=0.008):
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    alphas_cumprod = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0, 0.999)


def sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1, clamp_min=1e-5):
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    alphas_cumprod = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0, 0.999)


# ----------------------------------------------------------------------------- GaussianDiffusion (:540-867)
class GaussianDiffusion(_EpochOnApply, nn.Module):
    def __init__(self, model, *, image_size, timesteps=1000, sampling_timesteps=None, loss_type='l1',
                 objective='pred_noise', beta_schedule='sigmoid', schedule_fn_kwargs=dict(), p2_loss_weight_gamma=0.,
                 p2_loss_weight_k=1, ddim_sampling_eta=0., auto_normalize=True):
        super().__init__()
        assert not (type(self) == GaussianDiffusion and model.channels != model.out_dim)
        assert not model.random_or_learned_sinusoidal_cond
        self.model = model
        self.channels = self.model.channels
        self.self_condition = self.model.self_condition
        self.image_size = image_size
        self.objective = objective
        assert objective in {'pred_noise', 'pred_x0', 'pred_v'}, 'objective must be either pred_noise, pred_x0 or pred_v'
        if beta_schedule == 'linear':
            beta_schedule_fn = linear_beta_schedule
        elif beta_schedule == 'cosine':
            beta_schedule_fn = cosine_beta_schedule
        elif beta_schedule == 'sigmoid':
            beta_schedule_fn = sigmoid_beta_schedule
        else:
            raise ValueError(f'unknown beta schedule {beta_schedule}')
        betas = beta_schedule_fn(timesteps, **schedule_fn_kwargs)
        alphas = 1. - betas
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        alphas_cumprod_prev = torch.nn.functional.pad(alphas_cumprod[:-1], (1, 0), value=1.)
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
        if hasattr(self.model, "enable_time_table"):
            self.model.enable_time_table(self.num_timesteps)          # no-grad U-Net calls gather their time path from a [T, .] table
        self.loss_type = loss_type
        self.sampling_timesteps = default(sampling_timesteps, timesteps)
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        register_buffer = lambda name, val: self.register_buffer(name, val.to(torch.float32))
        register_buffer('betas', betas)
        register_buffer('alphas_cumprod', alphas_cumprod)
        register_buffer('alphas_cumprod_prev', alphas_cumprod_prev)
        register_buffer('sqrt_alphas_cumprod', torch.sqrt(alphas_cumprod))
        register_buffer('sqrt_one_minus_alphas_cumprod', torch.sqrt(1. - alphas_cumprod))
        register_buffer('log_one_minus_alphas_cumprod', torch.log(1. - alphas_cumprod))
        register_buffer('sqrt_recip_alphas_cumprod', torch.sqrt(1. / alphas_cumprod))
        register_buffer('sqrt_recipm1_alphas_cumprod', torch.sqrt(1. / alphas_cumprod - 1))
        posterior_variance = betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        register_buffer('posterior_variance', posterior_variance)
        register_buffer('posterior_log_variance_clipped', torch.log(posterior_variance.clamp(min=1e-20)))
        register_buffer('posterior_mean_coef1', betas * torch.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
        register_buffer('posterior_mean_coef2', (1. - alphas_cumprod_prev) * torch.sqrt(alphas) / (1. - alphas_cumprod))
        register_buffer('p2_loss_weight', (p2_loss_weight_k + alphas_cumprod / (1 - alphas_cumprod)) ** -p2_loss_weight_gamma)
        # kernel-side helper (not part of the reference's state): -sqrt(1/ac - 1) so x0 = a[t]*x + b[t]*eps is one launch
        self.register_buffer('_neg_sqrt_recipm1', -self.sqrt_recipm1_alphas_cumprod.clone(), persistent=False)
        # x0 = _x0_a[t] x_t - _x0_b[t] model_out for every objective (model_predictions, reference :674-694): the kernels that used to
        # take (sqrt_recip_ac, sqrt_recipm1_ac) take this pair - pred_noise: exactly those two (predict_start_from_noise :641-645);
        # pred_x0: (0, -1), i.e. the output itself; pred_v: (sqrt_ac, sqrt_one_minus_ac) (predict_start_from_v :659-663)
        if objective == 'pred_noise':
            x0_a, x0_b = self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod
        elif objective == 'pred_x0':
            x0_a, x0_b = torch.zeros_like(self.betas), -torch.ones_like(self.betas)
        else:
            x0_a, x0_b = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod
        self.register_buffer('_x0_a', x0_a.clone(), persistent=False)
        self.register_buffer('_x0_b', x0_b.clone(), persistent=False)
        self.register_buffer('_neg_sqrt_one_minus_ac', -self.sqrt_one_minus_alphas_cumprod.clone(), persistent=False)
        self.normalize = normalize_to_neg_one_to_one if auto_normalize else identity
        self.unnormalize = unnormalize_to_zero_to_one if auto_normalize else identity
        # The reverse step is ~150 small launches (launch-bound in eager mode): p_sample_loop replays ONE captured
        # hipGraph per timestep instead; set use_graph=False to force the eager loop.
        self.use_graph = True
        self._graphs = {}

    def _drop_stale_graphs(self) -> None:
        """A captured step reads the kernel-layout copies of the U-Net weights that existed at capture time.  When a weight has
        changed since (training between two sampling calls, load_state_dict, FusedAdam's epoch) the captured graphs are dropped
        and recaptured on next use; the schedule tables stay."""
        stamp = (K.param_epoch(),) + tuple((p.data_ptr(), p._version) for p in self.model.parameters())
        if stamp != getattr(self, "_graph_stamp", None):
            for k in [k for k in self._graphs if k[0] in ("step", "ddim", "vb")]:
                del self._graphs[k]
            self._graph_stamp = stamp

    # ---- hipGraph-captured reverse step: eps = Unet(x, t) -> fused p_sample in place -> t -= 1, state on the device
    def _step_inplace(self, xh: Tensor, tb: Tensor, gen: Optional[torch.Generator] = None) -> None:
        eps = self.model.forward_nhwc(xh, tb)
        noise = torch.randn_like(xh) if gen is None else torch.randn(xh.shape, device=xh.device, dtype=xh.dtype, generator=gen)
        K.p_sample(xh, eps, noise, tb, self._x0_a, self._x0_b,
                   self.posterior_mean_coef1, self.posterior_mean_coef2, self.posterior_log_variance_clipped,
                   want_x0=False, clip=True, out=xh)
        tb.sub_(1)

    def _captured_step(self, shape_nhwc, device):
        key = ("step", tuple(shape_nhwc), str(device))
        self._drop_stale_graphs()
        if key not in self._graphs:
            xh = torch.zeros(shape_nhwc, device=device)
            tb = torch.zeros(shape_nhwc[0], device=device, dtype=torch.long)
            # The step's noise comes from a generator of its own, registered with the graph: a replay writes the Philox
            # offset into a device tensor that belongs to the GENERATOR, so graphs sharing the default generator and
            # replayed on different streams (DiffusionPrior runs the latent parts concurrently) would overwrite each other's
            # offsets.  p_sample_loop reseeds it from the default generator, so torch.manual_seed still fixes the samples.
            gen = torch.Generator(device=device)
            warm = torch.cuda.Stream(device=device)           # warm-up off the default stream: packs weights, sets
            warm.wait_stream(torch.cuda.current_stream())     # kernel attributes, primes the allocator
            with torch.cuda.stream(warm):
                for _ in range(2):
                    tb.fill_(1)
                    self._step_inplace(xh, tb, gen)
            torch.cuda.current_stream().wait_stream(warm)
            g = torch.cuda.CUDAGraph()
            g.register_generator_state(gen)
            tb.fill_(1)
            with torch.cuda.graph(g):
                self._step_inplace(xh, tb, gen)
            self._graphs[key] = (g, xh, tb, gen)
        return self._graphs[key]

    # ---- closed-form pieces
    def predict_start_from_noise(self, x_t, t, noise):
        """sqrt(1/ac) x_t - sqrt(1/ac - 1) eps  (reference :641-645)."""
        xh, nh = K.to_nhwc(x_t).contiguous(), K.to_nhwc(noise).contiguous()
        return K.as_nchw(K.q_sample(xh, nh, t.contiguous(), self.sqrt_recip_alphas_cumprod, self._neg_sqrt_recipm1))

    def q_posterior(self, x_start, x_t, t):
        """(reference :665-672)"""
        xs, xt = K.to_nhwc(x_start).contiguous(), K.to_nhwc(x_t).contiguous()
        mean = K.as_nchw(K.q_sample(xs, xt, t.contiguous(), self.posterior_mean_coef1, self.posterior_mean_coef2))
        return mean, extract(self.posterior_variance, t, x_t.shape), extract(self.posterior_log_variance_clipped, t, x_t.shape)

    def predict_noise_from_start(self, x_t, t, x0):
        """(sqrt(1/ac) x_t - x0) / sqrt(1/ac - 1)  (reference :647-651)."""
        return K.as_nchw(K.noise_from_start(K.to_nhwc(x_t).contiguous(), K.to_nhwc(x0).contiguous(), t.contiguous(),
                                            self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod))

    def predict_v(self, x_start, t, noise):
        """sqrt(ac) noise - sqrt(1 - ac) x_start  (reference :653-657)."""
        return K.as_nchw(K.q_sample(K.to_nhwc(noise).contiguous(), K.to_nhwc(x_start).contiguous(), t.contiguous(),
                                    self.sqrt_alphas_cumprod, self._neg_sqrt_one_minus_ac))

    def predict_start_from_v(self, x_t, t, v):
        """sqrt(ac) x_t - sqrt(1 - ac) v  (reference :659-663)."""
        return K.as_nchw(K.q_sample(K.to_nhwc(x_t).contiguous(), K.to_nhwc(v).contiguous(), t.contiguous(),
                                    self.sqrt_alphas_cumprod, self._neg_sqrt_one_minus_ac))

    def _predictions_nhwc(self, xh: Tensor, t: Tensor, sch: Optional[Tensor], clip: bool, want_noise: bool = True):
        """model output -> (pred_noise or None, x_start, posterior mean) for the objective (reference :674-694 + q_posterior)."""
        out = self.model.forward_nhwc(xh, t, sch)
        mean, x0 = K.p_sample(xh, out, None, t, self._x0_a, self._x0_b, self.posterior_mean_coef1, self.posterior_mean_coef2,
                              self.posterior_log_variance_clipped, want_x0=True, clip=clip)
        if not want_noise:
            return None, x0, mean
        if self.objective == 'pred_noise':
            return out, x0, mean
        return K.noise_from_start(xh, x0, t, self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod), x0, mean

    def model_predictions(self, x, t, x_self_cond=None, clip_x_start=False):
        """(reference :674-694)"""
        xh = K.to_nhwc(x).contiguous()
        sch = None if x_self_cond is None else K.to_nhwc(x_self_cond).contiguous()
        eps, x0, _ = self._predictions_nhwc(xh, t.contiguous(), sch, clip_x_start)
        return ModelPrediction(K.as_nchw(eps), K.as_nchw(x0))

    def p_mean_variance(self, x, t, x_self_cond=None, clip_denoised=True):
        """(reference :696-704)"""
        xh = K.to_nhwc(x).contiguous()
        sch = None if x_self_cond is None else K.to_nhwc(x_self_cond).contiguous()
        _, x0, mean = self._predictions_nhwc(xh, t.contiguous(), sch, clip_denoised, want_noise=False)
        return (K.as_nchw(mean), extract(self.posterior_variance, t, x.shape),
                extract(self.posterior_log_variance_clipped, t, x.shape), K.as_nchw(x0))

    def _p_sample_nhwc(self, xh: Tensor, t: int, noise: Optional[Tensor], want_x0: bool, sch: Optional[Tensor] = None):
        b = xh.shape[0]
        tb = torch.full((b,), t, device=xh.device, dtype=torch.long)
        out = self.model.forward_nhwc(xh, tb, sch)
        if t > 0 and noise is None:
            noise = torch.randn_like(xh)
        return K.p_sample(xh, out, noise if t > 0 else None, tb, self._x0_a, self._x0_b, self.posterior_mean_coef1,
                          self.posterior_mean_coef2, self.posterior_log_variance_clipped, want_x0=want_x0, clip=True)

    @torch.no_grad()
    def p_sample(self, x, t: int, x_self_cond=None, noise: Optional[Tensor] = None):
        """One reverse step (reference :706-715); `noise` (logical NCHW) replaces the randn_like draw."""
        nh = None if noise is None else K.to_nhwc(noise).contiguous()
        sch = None if x_self_cond is None else K.to_nhwc(x_self_cond).contiguous()
        img, x0 = self._p_sample_nhwc(K.to_nhwc(x).contiguous(), t, nh, True, sch)
        return K.as_nchw(img), K.as_nchw(x0)

    @torch.no_grad()
    def p_sample_loop(self, shape, return_all_timesteps=False, x_T: Optional[Tensor] = None,
                      noises: Optional[Sequence[Tensor]] = None):
        """T sequential reverse steps (reference :717-733).  x_T / noises (one per step t = T-1..1, logical NCHW)
        replace the device RNG draws."""
        device = self.betas.device
        b, c, h, w = shape
        if self.use_graph and x_T is None and noises is None and not return_all_timesteps and self.num_timesteps > 1 \
                and not self.self_condition:
            it = self.p_sample_loop_iter(shape)
            try:
                while True:
                    next(it)
            except StopIteration as done:
                return done.value
        img = torch.randn((b, h, w, c), device=device) if x_T is None else K.to_nhwc(x_T.to(device)).contiguous()
        imgs = [img]
        x_start = None
        for i, t in enumerate(reversed(range(0, self.num_timesteps))):
            nz = None if noises is None or t == 0 else K.to_nhwc(noises[i].to(device)).contiguous()
            self_cond = x_start if self.self_condition else None               # (reference :726-728)
            img, x_start = self._p_sample_nhwc(img, t, nz, self.self_condition, self_cond)
            if return_all_timesteps:
                imgs.append(img)
        ret = K.as_nchw(img) if not return_all_timesteps else torch.stack([K.as_nchw(i) for i in imgs], dim=1)
        return self.unnormalize(ret)

    def p_sample_loop_iter(self, shape, x_start_nhwc: Optional[Tensor] = None, t_first: Optional[int] = None):
        """The captured-step loop as a generator: enqueues one timestep per next() on the CURRENT stream and returns the sample
        through StopIteration.  DiffusionPrior drives one of these per latent part, each under its own stream, in round-robin,
        so the parts' steps overlap on the GPU (a step is ~100 small kernels; one chain leaves most CUs idle).
        x_start_nhwc / t_first: start from a given x_t at timestep t_first instead of pure noise at T-1 (the reverse halves of
        sample_latent_given_start and interpolate)."""
        t_first = self.num_timesteps - 1 if t_first is None else int(t_first)
        with torch.no_grad():
            device = self.betas.device
            b, c, h, w = shape
            g, xh, tb, gen = self._captured_step((b, h, w, c), device)
            gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))       # from the default (CPU) generator: no device sync
            if x_start_nhwc is None:
                xh.normal_(generator=gen)
            else:
                xh.copy_(x_start_nhwc)
            tb.fill_(t_first)
        for _ in range(t_first):                              # t = t_first .. 1 (the graph decrements t itself)
            g.replay()
            yield
        with torch.no_grad():
            img, _ = self._p_sample_nhwc(xh, 0, None, False)   # t = 0: no noise
            return self.unnormalize(K.as_nchw(img))

    def reverse_from_iter(self, x_t: Tensor, t_first: int):
        """Reverse chain from a given x_t (logical NCHW) at timestep t_first down to 0: the captured loop when available (a
        generator, see p_sample_loop_iter), else the eager p_sample loop."""
        if self.use_graph and x_t.is_cuda and t_first >= 1 and not self.self_condition:
            return (yield from self.p_sample_loop_iter(tuple(x_t.shape), K.to_nhwc(x_t).contiguous(), t_first))
        with torch.no_grad():
            img = K.to_nhwc(x_t).contiguous()
            x_start = None
            for i in reversed(range(0, t_first + 1)):
                self_cond = x_start if self.self_condition else None
                img, x_start = self._p_sample_nhwc(img, i, None, self.self_condition, self_cond)
            return self.unnormalize(K.as_nchw(img))

    def sample_iter(self, batch_size=16):
        """Generator form of sample(): yields between timesteps where the loop can be interleaved with other chains (the
        hipGraph paths of p_sample_loop and ddim_sample); the eager loops run to completion inside the first next()."""
        shape = (batch_size, self.channels, self.image_size, self.image_size)
        if not self.is_ddim_sampling and self.use_graph and self.num_timesteps > 1 and not self.self_condition:
            return (yield from self.p_sample_loop_iter(shape))
        if self.is_ddim_sampling and self.use_graph and self.betas.is_cuda and not self.self_condition:
            return (yield from self.ddim_sample_iter(shape))
        return self.sample(batch_size)

    # ---- DDIM (the reference's yaml default: sampling_timesteps = 100 < T): the same captured-step machinery
    def _ddim_scalars(self, time: int, time_next: int, ac, sr, srm1):
        """Host-side fp32 scalars of one DDIM step (reference :749-773) as a 12-float row: (sr, srm1, sqrt(alpha_next), c, sigma, last,
        time, derive, x0_a, x0_b, 0, 0) - derive / x0_a / x0_b select the objective inside the kernel (nfdp_ddim_step_ex_f32)."""
        derive = 0.0 if self.objective == 'pred_noise' else 1.0
        x0_a, x0_b = float(self._x0_a[time]), float(self._x0_b[time])
        if time_next < 0:
            return [float(sr[time]), float(srm1[time]), 0.0, 0.0, 0.0, 1.0, float(time), derive, x0_a, x0_b, 0.0, 0.0]
        alpha, alpha_next = ac[time], ac[time_next]
        sigma = self.ddim_sampling_eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        cc = (1 - alpha_next - sigma ** 2).sqrt()
        return [float(sr[time]), float(srm1[time]), float(alpha_next.sqrt()), float(cc), float(sigma), 0.0, float(time), derive,
                x0_a, x0_b, 0.0, 0.0]

    def _ddim_table(self, device):
        """[S, 12] fp32 rows (_ddim_scalars), computed once on the host in fp32 from the schedule exactly as the eager loop below
        computes them."""
        key = ("ddim_table", str(device), self.sampling_timesteps, float(self.ddim_sampling_eta))
        if key not in self._graphs:
            total, S = self.num_timesteps, self.sampling_timesteps
            times = list(reversed(torch.linspace(-1, total - 1, steps=S + 1).int().tolist()))
            ac = self.alphas_cumprod.detach().cpu()
            sr, srm1 = self.sqrt_recip_alphas_cumprod.detach().cpu(), self.sqrt_recipm1_alphas_cumprod.detach().cpu()
            rows = [self._ddim_scalars(time, time_next, ac, sr, srm1) for time, time_next in zip(times[:-1], times[1:])]
            self._graphs[key] = torch.tensor(rows, dtype=torch.float32, device=device)
        return self._graphs[key]

    def _ddim_step_inplace(self, xh: Tensor, tb: Tensor, step: Tensor, table: Tensor, gen: torch.Generator) -> None:
        eps = self.model.forward_nhwc(xh, tb)
        noise = torch.randn(xh.shape, device=xh.device, dtype=xh.dtype, generator=gen)
        K.ddim_step_table(xh, eps, noise, table, step, tb)           # x in place; tb <- time of the next step
        step.add_(1)

    def _captured_ddim_step(self, shape_nhwc, device):
        key = ("ddim", tuple(shape_nhwc), str(device), self.sampling_timesteps, float(self.ddim_sampling_eta))
        self._drop_stale_graphs()
        if key not in self._graphs:
            table = self._ddim_table(device)
            xh = torch.zeros(shape_nhwc, device=device)
            tb = torch.zeros(shape_nhwc[0], device=device, dtype=torch.long)
            step = torch.zeros(1, device=device, dtype=torch.long)
            gen = torch.Generator(device=device)               # a generator per graph: see _captured_step
            warm = torch.cuda.Stream(device=device)
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                for _ in range(2):
                    step.zero_(); tb.fill_(int(table[0, 6].item()))
                    self._ddim_step_inplace(xh, tb, step, table, gen)
            torch.cuda.current_stream().wait_stream(warm)
            g = torch.cuda.CUDAGraph()
            g.register_generator_state(gen)
            step.zero_()
            with torch.cuda.graph(g):
                self._ddim_step_inplace(xh, tb, step, table, gen)
            self._graphs[key] = (g, xh, tb, step, gen, int(table[0, 6].item()))
        return self._graphs[key]

    def ddim_sample_iter(self, shape):
        """The DDIM loop on the captured step as a generator (one sampling step per next(), the sample through StopIteration):
        the U-Net's timestep and the step's scalars come from device tables indexed by a device-side counter."""
        with torch.no_grad():
            device = self.betas.device
            b, c, h, w = shape
            g, xh, tb, step, gen, t0 = self._captured_ddim_step((b, h, w, c), device)
            gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))
            xh.normal_(generator=gen)
            tb.fill_(t0)
            step.zero_()
        for _ in range(self.sampling_timesteps):
            g.replay()
            yield
        with torch.no_grad():
            return self.unnormalize(K.as_nchw(xh.clone()))

    @torch.no_grad()
    def ddim_sample(self, shape, return_all_timesteps=False, x_T: Optional[Tensor] = None,
                    noises: Optional[Sequence[Tensor]] = None):
        """(reference :735-777); the per-step scalars are computed in fp32 on the host from the schedule."""
        if self.use_graph and x_T is None and noises is None and not return_all_timesteps and self.betas.is_cuda \
                and not self.self_condition:
            it = self.ddim_sample_iter(shape)
            try:
                while True:
                    next(it)
            except StopIteration as done:
                return done.value
        device, total, S, eta = self.betas.device, self.num_timesteps, self.sampling_timesteps, self.ddim_sampling_eta
        times = torch.linspace(-1, total - 1, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        time_pairs = list(zip(times[:-1], times[1:]))
        b, c, h, w = shape
        img = torch.randn((b, h, w, c), device=device) if x_T is None else K.to_nhwc(x_T.to(device)).contiguous()
        imgs = [img]
        ac = self.alphas_cumprod.detach().cpu()
        sr, srm1 = self.sqrt_recip_alphas_cumprod.detach().cpu(), self.sqrt_recipm1_alphas_cumprod.detach().cpu()
        j = 0
        x_start = None
        for time, time_next in time_pairs:
            tb = torch.full((b,), time, device=device, dtype=torch.long)
            self_cond = x_start if self.self_condition else None                # (reference :745)
            out = self.model.forward_nhwc(img, tb, self_cond)
            row = self._ddim_scalars(time, time_next, ac, sr, srm1)
            nz = None
            if time_next >= 0:
                nz = torch.randn_like(img) if noises is None else K.to_nhwc(noises[j].to(device)).contiguous()
                j += 1
            img, x_start = K.ddim_step(img, out, nz, row[0], row[1], row[2], row[3], row[4], last=time_next < 0,
                                       x0_coef=(row[8], row[9]), derive_noise=row[7] != 0.0, want_x0=True)
            if return_all_timesteps:
                imgs.append(img)
        ret = K.as_nchw(img) if not return_all_timesteps else torch.stack([K.as_nchw(i) for i in imgs], dim=1)
        return self.unnormalize(ret)

    @torch.no_grad()
    def sample(self, batch_size=16, return_all_timesteps=False):
        """(reference :779-783)"""
        image_size, channels = self.image_size, self.channels
        sample_fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        return sample_fn((batch_size, channels, image_size, image_size), return_all_timesteps=return_all_timesteps)

    @torch.no_grad()
    def interpolate(self, x1=None, x2=None, t=None, lam=0.5):
        """(reference :785-804)"""
        b, device = x1.shape[0], x1.device
        t = default(t, self.num_timesteps - 1)
        assert x1.shape == x2.shape
        tb = torch.full((b,), t, device=device, dtype=torch.long)
        xt1, xt2 = self.q_sample(x1, t=tb), self.q_sample(x2, t=tb)
        img = (1 - lam) * xt1 + lam * xt2
        if t < 1:
            return img
        it = self.reverse_from_iter(img, t - 1)               # i = t-1 .. 0, as the reference's loop
        try:
            while True:
                next(it)
        except StopIteration as done:
            return done.value

    def q_sample(self, x_start, t, noise=None):
        """(reference :806-812)"""
        noise = default(noise, lambda: torch.randn_like(x_start))
        out = K.q_sample(K.to_nhwc(x_start).contiguous(), K.to_nhwc(noise).contiguous(), t.contiguous(),
                         self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)
        return K.as_nchw(out)

    @property
    def loss_fn(self):
        """Name check only: the reduction itself is the nfdp_diffusion_loss_f32 kernel  (reference :814-821)."""
        if self.loss_type in ('l1', 'l2'):
            return self.loss_type
        raise ValueError(f'invalid loss type {self.loss_type}')

    def p_losses(self, x_start, t, noise=None):
        """mean_b(mean_chw(|eps_hat - eps|^p) * p2w[t])  (reference :823-859)."""
        kind = self.loss_fn
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.model.parameters()):
            from . import training                     # differentiable path: autograd Functions over the HIP kernels
            return training.p_losses(self, x_start, t, noise)
        noise = default(noise, lambda: torch.randn_like(x_start))
        x0h, nh = K.to_nhwc(x_start).contiguous(), K.to_nhwc(noise).contiguous()
        t = t.contiguous()
        xh = K.q_sample(x0h, nh, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)
        sch = self._maybe_self_cond(xh, t)
        out = self.model.forward_nhwc(xh, t, sch)
        return K.diffusion_loss(out, self._loss_target_nhwc(x0h, nh, t), t, self.p2_loss_weight, l1=(kind == 'l1'))

    def _maybe_self_cond(self, xh: Tensor, t: Tensor) -> Optional[Tensor]:
        """Self-conditioning during training (reference :832-841): half of the time (python's random(), as the reference) the model's
        own x_start prediction, computed without gradient, is fed back as conditioning."""
        if not self.self_condition:
            return None
        if torch.cuda.is_current_stream_capturing():
            # the coin is drawn on the HOST: a hipGraph would freeze this step's outcome into every replay (always / never
            # self-conditioned).  Train self-conditioned models with eager steps, or capture two graphs and choose per step.
            raise RuntimeError("GaussianDiffusion(self_condition=True).p_losses draws the self-conditioning coin on the host "
                               "(reference :832) and cannot be captured in a hipGraph")
        if not _random() < 0.5:
            return None
        with torch.no_grad():
            _, x0, _ = self._predictions_nhwc(xh.detach(), t, None, False, want_noise=False)
        return x0.detach()

    def _loss_target_nhwc(self, x0h: Tensor, nh: Tensor, t: Tensor) -> Tensor:
        """What the model output is compared with (reference :847-855): the noise, x_start, or v = sqrt(ac) noise - sqrt(1-ac) x_start."""
        if self.objective == 'pred_noise':
            return nh
        if self.objective == 'pred_x0':
            return x0h.detach()
        return K.q_sample(nh, x0h.detach().contiguous(), t, self.sqrt_alphas_cumprod, self._neg_sqrt_one_minus_ac)

    def forward(self, img, *args, **kwargs):
        """(reference :861-867)"""
        b, c, h, w = img.shape
        assert h == self.image_size and w == self.image_size, f'height and width of image must be {self.image_size}'
        t = torch.randint(0, self.num_timesteps, (b,), device=img.device).long()
        img = self.normalize(img)
        return self.p_losses(img, t, *args, **kwargs)

    # ---- variational bound / NLL evaluation  (reference :869-958)
    def q_mean_variance(self, x_start, t):
        """q(x_t | x_0): (mean, variance, log_variance)  (reference :920-930)."""
        xh = K.to_nhwc(x_start).contiguous()
        mean = K.as_nchw(K.q_sample(xh, xh, t.contiguous(), self.sqrt_alphas_cumprod, torch.zeros_like(self.betas)))
        return (mean, extract(1.0 - self.alphas_cumprod, t, x_start.shape),
                extract(self.log_one_minus_alphas_cumprod, t, x_start.shape))

    def _prior_bpd(self, x_start):
        """mean_chw KL(q(x_T|x_0) || N(0,I)) per sample, one launch  (reference :905-918)."""
        return K.prior_kl(K.to_nhwc(x_start).contiguous(), self.sqrt_alphas_cumprod, self.log_one_minus_alphas_cumprod,
                          self.num_timesteps - 1)

    def _vb_term_nhwc(self, x0h, xth, t, clip_denoised=True, total=None, sch=None):
        eps = self.model.forward_nhwc(xth, t, sch)
        return K.vb_term(x0h, xth, eps, t, self._x0_a, self._x0_b,
                         self.posterior_mean_coef1, self.posterior_mean_coef2, self.posterior_log_variance_clipped,
                         clip=clip_denoised, total=total)

    @torch.no_grad()
    def _vb_terms_bpd(self, x_start, x_t, t, x_self_cond=None, clip_denoised=True, model_kwargs=None):
        """One term of the bound per sample: decoder NLL where t == 0, KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) elsewhere
        (reference :932-958).  U-Net forward + one fused launch (posterior means, normal_kl, tanh-CDF likelihood, mean)."""
        sch = None if x_self_cond is None else K.to_nhwc(x_self_cond).contiguous()
        return self._vb_term_nhwc(K.to_nhwc(x_start).contiguous(), K.to_nhwc(x_t).contiguous(), t.contiguous(), clip_denoised, sch=sch)

    def _vb_step_inplace(self, x0h: Tensor, tb: Tensor, total: Tensor, gen: Optional[torch.Generator] = None) -> None:
        noise = torch.randn_like(x0h) if gen is None else torch.randn(x0h.shape, device=x0h.device, dtype=x0h.dtype, generator=gen)
        xth = K.q_sample(x0h, noise, tb, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)
        self._vb_term_nhwc(x0h, xth, tb, True, total)
        tb.sub_(1)

    def _captured_vb_step(self, shape_nhwc, device):
        key = ("vb", tuple(shape_nhwc), str(device))
        self._drop_stale_graphs()
        if key not in self._graphs:
            x0h = torch.zeros(shape_nhwc, device=device)
            tb = torch.zeros(shape_nhwc[0], device=device, dtype=torch.long)
            total = torch.zeros(shape_nhwc[0], device=device, dtype=torch.float64)
            gen = torch.Generator(device=device)               # a generator per graph: see _captured_step
            warm = torch.cuda.Stream(device=device)
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                for _ in range(2):
                    tb.fill_(1)
                    self._vb_step_inplace(x0h, tb, total, gen)
            torch.cuda.current_stream().wait_stream(warm)
            g = torch.cuda.CUDAGraph()
            g.register_generator_state(gen)
            tb.fill_(1)
            with torch.cuda.graph(g):
                self._vb_step_inplace(x0h, tb, total, gen)
            self._graphs[key] = (g, x0h, tb, total, gen)
        return self._graphs[key]

    def calc_neg_log_likelihood_iter(self, x_start):
        """The captured variational-bound loop as a generator (one timestep per next(), the bound through StopIteration), so
        DiffusionPrior can interleave the latent parts' loops on separate streams like the sampling chains."""
        with torch.no_grad():
            x0h = K.to_nhwc(x_start).contiguous()
            g, x0s, tb, total, gen = self._captured_vb_step(tuple(x0h.shape), x0h.device)
            gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))
            x0s.copy_(x0h)
            total.zero_()
            tb.fill_(self.num_timesteps - 1)
        for _ in range(self.num_timesteps):
            g.replay()
            yield
        with torch.no_grad():
            return total.float() + self._prior_bpd(x_start)

    @torch.no_grad()
    def calc_neg_log_likelihood_loop(self, x_start, x_self_cond=None, clip_denoised=True,
                                     noises: Optional[Sequence[Tensor]] = None):
        """Whole variational bound per sample: sum over t = T-1..0 of the `_vb_terms_bpd` term at x_t ~ q(x_t|x_0), plus the
        prior term (reference :869-903; like the reference, `clip_denoised` is accepted but the terms always clip).
        The T terms are accumulated on the device in fp64 (the reference stacks them and sums in fp32); with device RNG
        one timestep (q_sample -> U-Net -> fused term -> t -= 1) is a hipGraph replayed T times.  `noises` (one per
        step, logical NCHW) replaces the randn_like draws."""
        sch = None if x_self_cond is None else K.to_nhwc(x_self_cond).contiguous()
        if self.use_graph and noises is None and sch is None:
            it = self.calc_neg_log_likelihood_iter(x_start)
            try:
                while True:
                    next(it)
            except StopIteration as done:
                return done.value
        x0h = K.to_nhwc(x_start).contiguous()
        b = x0h.shape[0]
        total = torch.zeros(b, device=x0h.device, dtype=torch.float64)
        for i, t in enumerate(reversed(range(self.num_timesteps))):
            tb = torch.full((b,), t, device=x0h.device, dtype=torch.long)
            nz = torch.randn_like(x0h) if noises is None else K.to_nhwc(noises[i].to(x0h.device)).contiguous()
            xth = K.q_sample(x0h, nz, tb, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)
            self._vb_term_nhwc(x0h, xth, tb, True, total, sch=sch)                 # (the reference passes the same x_self_cond to every term)
        return total.float() + self._prior_bpd(x_start)
