"""Flow helper modules + NLL math — mirrors reference normalizing_flow/utils.py (hot-path symbols only:
ZeroConv2d :17-44, Conv2dActNorm :47-69, coupling_network :72-90, calculate_output_shapes :93-117,
init_optimizer :120-137, preprocess_batch :175-196, postprocess_batch :199-210, calculate_loss :244-256,
initialize_with_zeros :259-272, data_dependent_nf_initialization :275-292, get_item :295-300).

All convolutions run as MFMA implicit GEMMs (csrc/conv.hip) on NHWC tensors with the ActNorm / exp(3*logs)
scaling and the ReLU folded into the epilogue.  Modules hold parameters in the reference's layout and under
the reference's state-dict names; kernel-layout copies (packed weights) are cached per parameter version.
"""
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

import nfdp_hip as K
from .base import default_device

LatentDim = Tuple[int, int, int]  # Channels, height, width.


class _PackedCache:
    """Kernel-layout copy of a conv weight, rebuilt when the parameter (version / storage / mode) changes."""

    def __init__(self):
        self._key = None
        self._packed = None

    def get(self, weight: Tensor, compute: int) -> Tensor:
        key = (weight.data_ptr(), weight._version, compute, weight.device, K.param_epoch())
        if key != self._key:
            self._packed = K.conv_pack(weight.detach(), compute)
            self._key = key
        return self._packed


    def get_taps(self, weight: Tensor, compute: int) -> Tensor:
        """Packed [9*O, I] weight of the tap GEMM (K.conv3x3_narrow)."""
        key = (weight.data_ptr(), weight._version, compute, weight.device, K.param_epoch())
        if key != getattr(self, "_key_taps", None):
            self._packed_taps = K.pack_taps(weight, compute)
            self._key_taps = key
        return self._packed_taps


class _ConvWeights(nn.Module):
    """Parameter container with nn.Conv2d's names, shapes and default init (never calls a torch conv)."""

    def __init__(self, in_channels: int, out_channels: int, k: int, bias: bool):
        super().__init__()
        self.in_channels, self.out_channels, self.k = in_channels, out_channels, k
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        bound = 1.0 / math.sqrt(in_channels * k * k)          # == kaiming_uniform_(a=sqrt(5)) of nn.Conv2d
        nn.init.uniform_(self.weight, -bound, bound)
        if bias:
            nn.init.uniform_(self.bias, -bound, bound)
        self._cache = _PackedCache()

    def packed(self, compute: int) -> Tensor:
        return self._cache.get(self.weight, compute)


class ZeroConv2d(nn.Module):
    """Zero-initialised 3x3 conv whose output is scaled by exp(logs*3)  (reference utils.py:17-44).

    State-dict keys: weight [O,I,3,3], bias [O], logs [1,O,1,1].
    """

    def __init__(self, in_channels: int, out_channels: int, filter_size: int = 3, stride: int = 1, padding: int = 0,
                 logscale: float = 3.):
        super().__init__()
        if stride != 1 or padding != (filter_size - 1) // 2:
            raise ValueError("the HIP conv implements stride 1 / 'same' padding (all the reference's call sites)")
        self.in_channels, self.out_channels, self.filter_size = in_channels, out_channels, filter_size
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, filter_size, filter_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.register_parameter("logs", nn.Parameter(torch.zeros(1, out_channels, 1, 1)))
        self.logscale_factor = logscale
        self._cache = _PackedCache()

    def taps_nhwc(self, xh: Tensor, compute: Optional[str] = None) -> Optional[Tensor]:
        """The tap contributions T [B,H,W,9*Cout] of this conv (its pointwise-GEMM half) where forward_nhwc would take the tap form,
        else None: AffineCoupling hands T to K.tap_coupling, which adds the taps up and applies the coupling in one pass."""
        c = K.compute_enum(compute)
        if xh.dtype == torch.bfloat16 and K.use_tap_conv(self.in_channels, self.out_channels, self.filter_size, c):
            return K.conv3x3_narrow(xh, self.weight, compute=c, packed_taps=self._cache.get_taps(self.weight, c), return_taps=True)
        return None

    def forward_nhwc(self, xh: Tensor, compute: Optional[str] = None) -> Tensor:
        c = K.compute_enum(compute)
        if xh.dtype == torch.bfloat16 and K.use_tap_conv(self.in_channels, self.out_channels, self.filter_size, c):
            # 512 -> C with C small: as an implicit GEMM the im2col staging dominates (10-70 TFLOP/s); computed instead as a
            # pointwise GEMM to the 9*C tap contributions + tap shift-sum (the decomposition of the fused flow-step kernel)
            return K.conv3x3_narrow(xh, self.weight, add=self.bias, mul=self.logs.view(-1), mul_exp=self.logscale_factor,
                                    relu=False, compute=c, packed_taps=self._cache.get_taps(self.weight, c))
        return K.conv2d(xh, self._cache.get(self.weight, c), self.out_channels, self.filter_size, add=self.bias,
                        mul=self.logs.view(-1), mul_exp=self.logscale_factor, relu=False, compute=c, out_dtype=K.F32)

    def forward(self, x: Tensor) -> Tensor:
        return K.as_nchw(self.forward_nhwc(K.to_nhwc(x)))


class Conv2dActNorm(nn.Module):
    """Bias-free conv followed by ActNorm  (reference utils.py:47-69); keys _Conv2dActNorm__conv.weight,
    _Conv2dActNorm__actnorm.{scale,bias,is_initialized}."""

    def __init__(self, in_channels: int, out_channels: int, filter_size: int, stride: int = 1, padding: int = None):
        super().__init__()
        from .transforms import ActNorm
        if stride != 1:
            raise ValueError("stride 1 only")
        self.filter_size = filter_size
        self.__conv = _ConvWeights(in_channels, out_channels, filter_size, bias=False)
        self.__actnorm = ActNorm(out_channels)

    @property
    def actnorm(self):
        return self.__actnorm

    @property
    def conv(self):
        return self.__conv

    def forward_nhwc(self, xh: Tensor, relu: bool = False, compute: Optional[str] = None, out_dtype: int = K.F32) -> Tensor:
        c = K.compute_enum(compute)
        an, cv = self.__actnorm, self.__conv
        packed = cv.packed(c)
        if not an.initialized():
            # data-dependent init (transforms.py:74-78) on the raw conv output, then the fused pass below
            raw = K.conv2d(xh, packed, cv.out_channels, self.filter_size, compute=c, out_dtype=K.F32)
            an.init_from(raw)
        k = self.filter_size
        if xh.dtype == torch.float32 and xh.shape[0] * xh.shape[1] * xh.shape[2] >= 4096 and \
                K.use_im2col_conv(cv.in_channels, cv.out_channels, k, c):
            # C/2 -> 512 on a narrow fp32 tensor: explicit im2col (9C wide) + the LDS-DMA pointwise GEMM; the packed 3x3
            # weight is the 1x1 weight for the im2col rows (same bf16 operands and summation order as the implicit GEMM)
            xh, k = K.im2col3x3(xh), 1
        return K.conv2d(xh, packed, cv.out_channels, k, add=an.bias.view(-1), mul=an.scale.view(-1),
                        mul_exp=1.0, relu=relu, compute=c, out_dtype=out_dtype)

    def forward(self, x: Tensor) -> Tensor:
        return K.as_nchw(self.forward_nhwc(K.to_nhwc(x)))


class CouplingNet(nn.Sequential):
    """The coupling network: conv3x3+ActNorm+ReLU -> conv1x1+ActNorm+ReLU -> ZeroConv3x3 as three MFMA launches
    with fused epilogues (hidden activations bf16 in bf16 mode), or — see fused_packed() — folded with the
    coupling epilogue into ONE launch.  Indexing/keys identical to the reference's nn.Sequential (net.0, net.2,
    net.4; ReLUs at 1 and 3)."""

    def fused_ready(self) -> bool:
        return self[0].actnorm.initialized() and self[2].actnorm.initialized()

    def fused_packed(self, kind: int = 1):
        """MFMA-register-image pack of all nine parameter tensors for nfdp_coupling_fused_bf16, rebuilt when any
        of them changes.  None while a hidden ActNorm still awaits its data-dependent init.  kind: the fragment layout the consuming
        kernel reads (K.coupling_fused_image_kind for the whole-image kernels; the tiled path reads kind 1); one cached image per kind."""
        c1, c2, zc = self[0], self[2], self[4]
        if not (c1.actnorm.initialized() and c2.actnorm.initialized()):
            return None
        ts = (c1.conv.weight, c1.actnorm.scale, c1.actnorm.bias, c2.conv.weight, c2.actnorm.scale, c2.actnorm.bias,
              zc.weight, zc.bias, zc.logs)
        key = tuple((t.data_ptr(), t._version) for t in ts) + (K.param_epoch(),)
        cache = self.__dict__.setdefault("_fused_images", {})
        hit = cache.get(kind)
        if hit is None or hit[0] != key:
            img = K.coupling_fused_pack(*[t.detach().reshape(-1) if t.dim() != 4 else t.detach() for t in ts], fold=kind,
                                        out=None if hit is None else hit[1])
            cache[kind] = hit = (key, img)
        return hit[1]

    def mlp_f32_packed(self):
        """The fragment-ordered fp32 image of the network for nfdp_coupling_mlp_f32 (compute="f32"), rebuilt when a parameter changes;
        None while a hidden ActNorm still awaits its data-dependent init."""
        c1, c2, zc = self[0], self[2], self[4]
        if not (c1.actnorm.initialized() and c2.actnorm.initialized()):
            return None
        ts = (c1.conv.weight, c1.actnorm.scale, c1.actnorm.bias, c2.conv.weight, c2.actnorm.scale, c2.actnorm.bias, zc.weight)
        key = tuple((t.data_ptr(), t._version) for t in ts) + (K.param_epoch(),)
        if getattr(self, "_mlp_f32_key", None) != key:
            self._mlp_f32 = K.coupling_mlp_f32_pack(*[t.detach().reshape(-1) if t.dim() != 4 else t.detach().contiguous() for t in ts],
                                                    out=getattr(self, "_mlp_f32", None))
            self._mlp_f32_key = key
        return self._mlp_f32

    def _pack_params(self):
        c1, c2, zc = self[0], self[2], self[4]
        return (c1.conv.weight, c1.actnorm.scale, c1.actnorm.bias, c2.conv.weight, c2.actnorm.scale, c2.actnorm.bias,
                zc.weight, zc.bias, zc.logs)

    def fused_pack_tensors(self):
        """The nine tensors the packers read, detached (vectors flattened)."""
        return tuple(t.detach().reshape(-1) if t.dim() != 4 else t.detach() for t in self._pack_params())

    def fused_train_key(self):
        return tuple((t.data_ptr(), t._version) for t in self._pack_params()) + (K.param_epoch(),)

    def fused_packed_train(self):
        """The image of the training forward (nfdp_level_fused_train_bf16: plain weights, scales applied in the epilogues), rebuilt
        into the same buffer when a parameter changes - every step while training (a whole level's images in one launch:
        training._level_fused)."""
        key = self.fused_train_key()
        if getattr(self, "_fused_train_key", None) != key:
            self._fused_train = K.coupling_fused_pack(*self.fused_pack_tensors(), fold=False, out=getattr(self, "_fused_train", None))
            self._fused_train_key = key
        return self._fused_train

    def forward_nhwc(self, xh: Tensor, compute: Optional[str] = None, return_taps: bool = False) -> Tensor:
        """return_taps: (T, None) when the output conv runs in its tap form (T = its tap contributions, for K.tap_coupling), else
        (None, h)."""
        c = K.compute_enum(compute)
        hid = K.BF16 if c == K.BF16 else K.F32
        h = self[0].forward_nhwc(xh, relu=True, compute=compute, out_dtype=hid)
        h = self[2].forward_nhwc(h, relu=True, compute=compute, out_dtype=hid)
        if return_taps:
            t = self[4].taps_nhwc(h, compute=compute)
            return (t, None) if t is not None else (None, self[4].forward_nhwc(h, compute=compute))
        return self[4].forward_nhwc(h, compute=compute)

    def forward(self, x: Tensor) -> Tensor:
        return K.as_nchw(self.forward_nhwc(K.to_nhwc(x)))


def coupling_network(in_channels: int, n_features: int = 512, out_channels: int = None):
    """(reference utils.py:72-90)"""
    return CouplingNet(
        Conv2dActNorm(in_channels, n_features, 3, padding=1),
        nn.ReLU(inplace=True),
        Conv2dActNorm(n_features, n_features, 1, padding=0),
        nn.ReLU(inplace=True),
        ZeroConv2d(n_features, out_channels or in_channels, padding=1),
    )


def calculate_output_shapes(L: int, in_channels: int, size: int) -> List[LatentDim]:
    """Latent part shapes, e.g. L=3,C=3,size=32 -> [(6,16,16),(12,8,8),(48,4,4)]  (reference utils.py:93-117)."""
    z_shapes = []
    for _ in range(L - 1):
        if size % 2 != 0:
            raise ValueError("The input dimension is not divisible by 2!")
        in_channels *= 2
        size //= 2
        z_shapes.append((in_channels, size, size))
    z_shapes.append((in_channels * 4, size // 2, size // 2))
    return z_shapes


def init_optimizer(name: str, params, lr: float) -> torch.optim.Optimizer:
    """(reference utils.py:120-137)"""
    if name == "adam":
        from torch.optim import Adam as Optimizer
    elif name == "adamw":
        from torch.optim import AdamW as Optimizer
    else:
        raise ValueError("Unknown optimizer")
    return Optimizer(params, lr=lr if lr else params[0]["lr"])


@torch.no_grad()
def preprocess_batch(batch: Tensor, n_bits: int, n_bins: int) -> Tensor:
    """[0,1] images -> n_bits quantised, centred  (reference utils.py:175-196).  uint8 input is also accepted
    (then batch/255 is applied in-kernel exactly as torchvision's ToTensor would have)."""
    if batch.dtype == torch.uint8:
        return K.as_nchw(K.preprocess_u8(batch.contiguous(), None, n_bits))
    return K.preprocess_f32(batch, None, n_bits)


@torch.no_grad()
def postprocess_batch(batch: Tensor, n_bins: int) -> Tensor:
    """-> uint8 on the host  (reference utils.py:199-210)."""
    return K.postprocess_u8(K.to_nhwc(batch).contiguous(), n_bins).to("cpu")


def calculate_loss(log_likelihood: Tensor, n_bins: float, n_pixel: float):
    """Bits per dimension, mean over the batch  (reference utils.py:244-256)."""
    if torch.is_grad_enabled() and log_likelihood.requires_grad:
        from . import training
        return training.bits_per_dim(log_likelihood, n_bins, n_pixel)
    ll = log_likelihood if log_likelihood.dtype == torch.float64 else log_likelihood.double()
    s = K.bpd_sum(ll.contiguous(), None, n_bins, n_pixel)
    return s[0] / s[1]


def initialize_with_zeros(n: int, batch_size: int, device: torch.device):
    """n fp64 zero accumulators  (reference utils.py:259-272)."""
    if n == 1:
        return torch.zeros(batch_size, device=device, dtype=torch.float64)
    return (torch.zeros(batch_size, device=device, dtype=torch.float64) for _ in range(n))


@torch.no_grad()
def data_dependent_nf_initialization(flow, dataloader, device: torch.device, n_bits: int, n_bins: int) -> None:
    """Runs one batch through the flow so the ActNorms initialise  (reference utils.py:275-292)."""
    flow.eval()
    sample = next(iter(dataloader))
    batch = sample[0].to(device) if isinstance(sample, list) else sample.to(device)
    batch = preprocess_batch(batch, n_bits, n_bins)
    log_likelihood, logp = initialize_with_zeros(2, batch.size(0), device)
    _ = flow.transform(batch + torch.rand_like(batch) / n_bins, log_likelihood, logp)


def get_item(sequence: Sequence, index: int):
    """sequence[index] or None  (reference utils.py:295-300)."""
    try:
        return sequence[index]
    except IndexError:
        return None
