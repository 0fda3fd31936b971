"""Flow primitives on the MI355X kernel library — mirrors reference normalizing_flow/transforms.py:
IdentityTransform :14-25, ActNorm :28-94, InvConv2d :97-145, AffineCoupling :148-201, Squeeze :204-239,
Split :242-309.  Same constructor arguments, method names, return tuples (accumulators updated in place and
returned) and state-dict keys.

Public methods take/return logical-NCHW tensors like the reference; outputs are channels_last-strided views
of the NHWC buffers the kernels work on (zero-copy).  `*_nhwc` methods are the internal fast path used by
StepFlow / GlowBlock / Glow so a whole flow runs without layout round trips.
Optional keyword extensions (never required): `eps=` injects the N(0,1) noise a sampling branch would draw.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

import nfdp_hip as K
from .base import Transform, Prior, acc_in, acc_out
from .prior import IsotropicGaussian
from .utils import coupling_network, ZeroConv2d


class IdentityTransform(Transform):
    """(reference transforms.py:14-25)"""

    def transform(self, x, log_det_jac, logp):
        return x, log_det_jac, logp

    def invert(self, y):
        return y


class ActNorm(Transform):
    """Per-channel affine layer with data-dependent initialisation  (reference transforms.py:28-94)."""

    def __init__(self, in_channels: int = 3):
        super().__init__()
        self.scale = nn.Parameter(torch.zeros(size=(in_channels, 1, 1), device=self.device))
        self.bias = nn.Parameter(torch.zeros(size=(in_channels, 1, 1), device=self.device))
        self.register_buffer("is_initialized", torch.tensor(0, dtype=torch.uint8, device=self.device))
        # host mirror of the flag (the reference syncs with .item() on every call): (buffer identity, buffer version, value).
        # copy_/fill_ on the buffer, load_state_dict, .to() and re-assigning the attribute all change the key; a write through
        # `.is_initialized.data` does not (it has its own version counter) - call reset_init_cache() after one.
        self._init_cache = None

    def reset_init_cache(self) -> None:
        self._init_cache = None

    def _load_from_state_dict(self, *args, **kwargs):
        self._init_cache = None
        return super()._load_from_state_dict(*args, **kwargs)

    def initialized(self) -> bool:
        buf = self.is_initialized
        key = (id(buf), buf.data_ptr(), buf._version)
        if self._init_cache is None or self._init_cache[0] != key:
            self._init_cache = (key, bool(buf.item() != 0))           # one sync per (re)load, not per call
        return self._init_cache[1]

    @torch.no_grad()
    def init_from(self, xh: Tensor) -> None:
        """scale = -log(std+1e-6), bias = -mean over (B,H,W)  (reference transforms.py:74-78)."""
        K.actnorm_init(xh, self.scale.view(-1), self.bias.view(-1))
        from nfdp_hip import dist as _dist
        if _dist._ACTNORM_SYNC["active"]:                       # nfdp_hip.dist.data_dependent_init(mode="global"): moments of all ranks
            _dist.actnorm_init_sync(self.scale.data, self.bias.data, xh.numel() // xh.shape[-1])
        self.is_initialized.fill_(1)
        buf = self.is_initialized
        self._init_cache = ((id(buf), buf.data_ptr(), buf._version), True)

    def transform_nhwc(self, xh: Tensor, ldj: Optional[Tensor]) -> Tensor:
        if not self.initialized():
            self.init_from(xh)
        y = K.actnorm_fwd(xh, self.scale.view(-1), self.bias.view(-1))
        if ldj is not None:
            K.ldj_add_scaled(ldj, self.scale.view(-1), self.scale.numel(), xh.shape[1] * xh.shape[2])
        return y

    def invert_nhwc(self, yh: Tensor) -> Tensor:
        return K.actnorm_inv(yh, self.scale.view(-1), self.bias.view(-1))

    def transform(self, x: Tensor, log_det_jac: Tensor, logp: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        w = acc_in(log_det_jac)
        y = self.transform_nhwc(K.to_nhwc(x).contiguous(), w)
        return K.as_nchw(y), acc_out(log_det_jac, w), logp

    def invert(self, y: Tensor) -> Tensor:
        return K.as_nchw(self.invert_nhwc(K.to_nhwc(y).contiguous()))


class InvConv2d(Transform):
    """Invertible 1x1 convolution  (reference transforms.py:97-145)."""

    def __init__(self, in_channels: int = 3):
        super().__init__()
        random_matrix = torch.randn(size=(in_channels, in_channels), dtype=torch.float32)
        w_init, _ = torch.linalg.qr(random_matrix)                       # host-side init only (transforms.py:112-114)
        # linalg.qr returns a column-major Q: force the row-major [out,in] layout the kernels (and OIHW) assume
        self.weight = nn.Parameter(w_init.contiguous().reshape(in_channels, in_channels, 1, 1).contiguous().to(self.device))

    def _w(self) -> Tensor:
        """Row-major [C,C] view of the parameter (re-densified in place if something re-strided it)."""
        if not self.weight.is_contiguous():
            self.weight.data = self.weight.data.contiguous()
        c = self.weight.shape[0]
        return self.weight.detach().view(c, c)

    def transform_nhwc(self, xh: Tensor, ldj: Optional[Tensor]) -> Tensor:
        if ldj is not None:
            lad, _ = K.invconv_prepare(self._w(), want_inverse=False)    # fp64 log|det| every call, as the reference
            K.ldj_add_scaled(ldj, lad, 1, xh.shape[1] * xh.shape[2])
        return K.invconv_apply(xh, self._w())

    def invert_nhwc(self, yh: Tensor) -> Tensor:
        _, winv = K.invconv_prepare(self._w(), want_inverse=True)
        return K.invconv_apply(yh, winv)

    def transform(self, x: Tensor, log_det_jac: Tensor, logp: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        w = acc_in(log_det_jac)
        y = self.transform_nhwc(K.to_nhwc(x).contiguous(), w)
        return K.as_nchw(y), acc_out(log_det_jac, w), logp

    def invert(self, y: Tensor) -> Tensor:
        return K.as_nchw(self.invert_nhwc(K.to_nhwc(y).contiguous()))


class AffineCoupling(Transform):
    """Affine coupling layer  (reference transforms.py:148-201).  `compute` ('f32' | 'bf16' | None = library
    default) selects the MFMA mode of the coupling network."""

    def __init__(self, in_channels: int = 2, n_features: int = 512, compute: Optional[str] = None, fused: bool = True):
        super().__init__()
        self.compute = compute
        self.fused = fused          # allow the one-launch fused kernel (bf16 mode, supported geometry)
        self.n_features = n_features
        self.net = coupling_network(in_channels=in_channels // 2, n_features=n_features,
                                    out_channels=in_channels).to(self.device)

    def _fused_pack(self, xh: Tensor):
        if not self.fused or K.compute_enum(self.compute) != K.BF16:
            return None
        _, h, w, c = xh.shape
        if not K.coupling_fused_infer_supported(h, w, c, self.n_features) or not self.net.fused_ready():
            return None
        return self.net.fused_packed(K.coupling_fused_image_kind(h, w, c))

    def _net_nhwc(self, xh: Tensor) -> Tensor:
        """NN(x_a) -> (log_s | t).  Images larger than the fused flow-step kernels' (which own whole 16x16 / 8x8 / 4x4 images) take
        the tiled form in bf16 mode: explicit im2col -> per-pixel GEMM chain with the hidden activations on chip -> tap shift-sum;
        otherwise the three convs one by one."""
        b, hh, ww, c = xh.shape
        if self.fused and K.compute_enum(self.compute) == K.BF16 and b * hh * ww >= 1024 and \
                K.coupling_mlp_supported(c, self.n_features) and self.net.fused_ready():
            zc = self.net[-1]
            kind = K.coupling_mlp_image_kind(xh[..., : c // 2], c)
            return K.coupling_net_tiled(xh[..., : c // 2], self.net.fused_packed(kind), c, zc.bias, zc.logs.view(-1), zc.logscale_factor, kind=kind)
        if self.fused and K.compute_enum(self.compute) == K.F32 and xh.dtype == torch.float32 and \
                K.coupling_mlp_f32_supported(c, self.n_features) and self.net.fused_ready():
            # exact fp32 with the hidden activations on chip (any image size): 3x3 gather + three fp32-MFMA GEMMs in one launch
            zc = self.net[-1]
            return K.coupling_net_tiled_f32(xh[..., : c // 2], self.net.mlp_f32_packed(), c, zc.bias, zc.logs.view(-1),
                                            zc.logscale_factor)
        return self.net.forward_nhwc(xh[..., : c // 2], compute=self.compute)

    def _net_taps_nhwc(self, xh: Tensor):
        """bf16 mode outside the whole-image kernels: the network up to the output conv's TAP CONTRIBUTIONS T [B,H,W,9C] (the
        shift-sum, the ZeroConv2d epilogue and the coupling then run as one pass, K.tap_coupling); None where the path does not apply."""
        if xh.dtype != torch.float32 or not self.net.fused_ready():
            return None
        b, hh, ww, c = xh.shape
        if K.compute_enum(self.compute) == K.F32:
            # exact-fp32 mode with split products (K.set_f32_coupling("bf16x3")): high- and low-piece images, three bf16 MFMAs per product
            if self.fused and K.coupling_mlp_x3_supported(xh[..., : c // 2], c, self.n_features):
                zc = self.net[-1]
                kind = K.coupling_mlp_image_kind(xh[..., : c // 2], c, split=True)
                return K.coupling_net_tiled(xh[..., : c // 2], self.net.fused_packed(kind), c, zc.bias, zc.logs.view(-1), zc.logscale_factor,
                                            return_taps=True, packed_lo=self.net.fused_packed(kind + 2), kind=kind)
            return None
        if K.compute_enum(self.compute) != K.BF16:
            return None
        if self.fused and b * hh * ww >= 1024 and K.coupling_mlp_supported(c, self.n_features):
            zc = self.net[-1]
            kind = K.coupling_mlp_image_kind(xh[..., : c // 2], c)
            return K.coupling_net_tiled(xh[..., : c // 2], self.net.fused_packed(kind), c, zc.bias, zc.logs.view(-1), zc.logscale_factor,
                                        return_taps=True, kind=kind)
        t, _ = self.net.forward_nhwc(xh[..., : c // 2], compute=self.compute, return_taps=True) if self._taps_ok() else (None, None)
        return t

    def _taps_ok(self) -> bool:
        zc = self.net[-1]
        return K.use_tap_conv(zc.in_channels, zc.out_channels, zc.filter_size, K.BF16)

    def transform_nhwc_(self, xh: Tensor, ldj: Optional[Tensor], ldj_const: Optional[Tensor] = None, next_head=None, ldj_defer=None):
        """In place on xh's second channel half.  ldj_const: device [2] fp32 added before the coupling sum.
        next_head: optional (scale [C], bias [C], W [C,C], logabsdet [1] | None) of the NEXT flow's ActNorm + 1x1 conv.  Where the layer
        runs as tap contributions + one pass (K.tap_coupling) that pass applies the head too and the call returns (xh, const) with the
        next flow's [2] log-det constants (None when ldj is None); otherwise xh alone - the caller then runs the head itself."""
        packed = self._fused_pack(xh)
        if packed is not None:
            K.coupling_fused(xh, packed, ldj, ldj_const, inverse=False)
            return xh
        t = self._net_taps_nhwc(xh)
        if t is not None:
            zc = self.net[-1]
            if next_head is not None and K.tap_coupling_head_supported(xh.shape[-1]):
                # (ldj_defer: the level loop's list - this launch's log-det sums are added to ldj by one launch at the end of the level)
                cst = K.tap_coupling_head(xh, t, zc.bias, zc.logs.view(-1), zc.logscale_factor, ldj, ldj_const, False,
                                          next_head[0], next_head[1], next_head[2], next_head[3], want_const=ldj is not None,
                                          defer=ldj_defer)
                return xh, cst
            K.tap_coupling(xh, t, zc.bias, zc.logs.view(-1), zc.logscale_factor, ldj, ldj_const, inverse=False)
            return xh
        h = self._net_nhwc(xh)
        K.coupling_fwd(xh, h, ldj, ldj_const)
        return xh

    def invert_nhwc_(self, yh: Tensor, head=None):
        """head: optional (scale [C], bias [C], W^-1 [C,C]) of THIS flow's ActNorm + 1x1 conv; where the layer runs as tap contributions +
        one pass, that pass applies the inverse head too and the call returns (yh, True); otherwise yh alone."""
        packed = self._fused_pack(yh)
        if packed is not None:
            K.coupling_fused(yh, packed, None, None, inverse=True)
            return yh
        t = self._net_taps_nhwc(yh)
        if t is not None:
            zc = self.net[-1]
            if head is not None and K.tap_coupling_head_supported(yh.shape[-1]):
                K.tap_coupling_head(yh, t, zc.bias, zc.logs.view(-1), zc.logscale_factor, None, None, True, head[0], head[1], head[2])
                return yh, True
            K.tap_coupling(yh, t, zc.bias, zc.logs.view(-1), zc.logscale_factor, None, None, inverse=True)
            return yh
        h = self._net_nhwc(yh)
        K.coupling_inv(yh, h)
        return yh

    def transform(self, x: Tensor, log_det_jac: Tensor, logp: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        w = acc_in(log_det_jac)
        y = self.transform_nhwc_(K.to_nhwc(x).contiguous().clone(), w)
        return K.as_nchw(y), acc_out(log_det_jac, w), logp

    def invert(self, y: Tensor) -> Tensor:
        return K.as_nchw(self.invert_nhwc_(K.to_nhwc(y).contiguous().clone()))


class Squeeze(Transform):
    """Space-to-depth by 2  (reference transforms.py:204-239); bit-exact index reshuffle."""

    def transform_nhwc(self, xh: Tensor) -> Tensor:
        return K.squeeze(xh)

    def invert_nhwc(self, yh: Tensor) -> Tensor:
        return K.unsqueeze(yh)

    def transform(self, x: Tensor, log_det_jac: Tensor, logp: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        return K.as_nchw(K.squeeze(K.to_nhwc(x))), log_det_jac, logp

    def invert(self, y: Tensor) -> Tensor:
        return K.as_nchw(K.unsqueeze(K.to_nhwc(y).contiguous()))


class Split(Transform):
    """Channel split with a learned Gaussian prior on the dropped half  (reference transforms.py:242-309)."""

    def __init__(self, in_channels, learn_prior_mean_logs: bool = True, compute: Optional[str] = None):
        super().__init__()
        self.compute = compute
        self.conv = ZeroConv2d(in_channels // 2, in_channels, padding=(3 - 1) // 2) if learn_prior_mean_logs else None
        if self.conv is not None:
            self.conv.to(self.device)

    def _mean_logs_nhwc(self, yh: Tensor) -> Tensor:
        """h = (mean | logs) [B,H,W,2*Cy]  (reference transforms.py:256-268; zeros when nothing is learned)."""
        if self.conv is not None:
            return self.conv.forward_nhwc(yh, compute=self.compute)
        b, h, w, c = yh.shape
        return torch.zeros((b, h, w, 2 * c), device=yh.device, dtype=torch.float32)

    def transform_nhwc(self, xh: Tensor, logp: Optional[Tensor], side: Optional["torch.cuda.Stream"] = None):
        """-> (y view, z contiguous copy); logp updated in place (skipped when None, transforms.py:287).
        side: a HIP stream for the prior term (the prior conv + log-density only feed logp, which nothing reads before the end of the
        forward pass): forked behind the work enqueued so far, NOT joined here - the caller joins it before logp is read."""
        c = xh.shape[-1]
        yv, zv = xh[..., : c // 2], xh[..., c // 2:]
        if logp is not None:
            if side is not None:
                main = torch.cuda.current_stream(xh.device)
                side.wait_stream(main)
                xh.record_stream(side)
                with torch.cuda.stream(side):
                    K.gauss_logp(zv, self._mean_logs_nhwc(yv), 0, logp)
            else:
                K.gauss_logp(zv, self._mean_logs_nhwc(yv), 0, logp)
        z = torch.empty(zv.shape, device=xh.device, dtype=torch.float32)
        K.copy_channels(zv, z)
        return yv, z

    def transform_squeeze_nhwc(self, xh: Tensor, logp: Optional[Tensor]):
        """Split.transform + the NEXT block's Squeeze in one launch (K.split_prior_squeeze; check K.split_prior_squeeze_supported
        first): -> (z contiguous, the kept half squeezed = the next level's buffer); logp updated in place (skipped when None)."""
        cv = self.conv
        return K.split_prior_squeeze(xh, None if cv is None else cv.weight, None if cv is None else cv.bias,
                                     None if cv is None else cv.logs.view(-1), logp)

    def invert_nhwc(self, yh: Tensor, zh: Optional[Tensor], temperature: float = 1.0, eps: Optional[Tensor] = None) -> Tensor:
        b, h, w, c = yh.shape
        out = torch.empty((b, h, w, 2 * c), device=yh.device, dtype=torch.float32)
        K.copy_channels(yh, out[..., :c])
        if zh is None:
            if eps is None:
                eps = torch.randn((b, h, w, c), device=yh.device, dtype=torch.float32)
            K.gauss_sample(self._mean_logs_nhwc(yh), 0, eps.contiguous(), out[..., c:], temperature)
        else:
            K.copy_channels(zh, out[..., c:])
        return out

    def transform(self, x: Tensor, log_det_jac: Tensor, logp: Tensor):
        w = acc_in(logp)
        yv, z = self.transform_nhwc(K.to_nhwc(x).contiguous(), w)
        return K.as_nchw(yv), log_det_jac, K.as_nchw(z), acc_out(logp, w)

    def invert(self, y: Tensor, inv_y_split: Tensor = None, temperature: float = 1.0, eps: Optional[Tensor] = None) -> Tensor:
        zh = None if inv_y_split is None else K.to_nhwc(inv_y_split)
        eh = None if eps is None else K.to_nhwc(eps).contiguous()
        return K.as_nchw(self.invert_nhwc(K.to_nhwc(y), zh, temperature, eh))
