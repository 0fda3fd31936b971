"""ctypes binding of the nfdp C ABI (include/nfdp.h) + thin torch-tensor wrappers.

torch is used here only as plumbing: device memory (torch.empty), the current HIP stream, and
tensor metadata.  Every arithmetic op on the hot path is a hand-written gfx950 kernel reached
through the C ABI.  There is NO CPU or eager-torch fallback: if the shared library is missing
or no GPU is visible, calls raise.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NFDP_LIB_PATH") or os.path.join(_HERE, "libnfdp_hip.so")   # override: profiling builds
HEADER_PATH = os.environ.get("NFDP_HEADER_PATH") or os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "nfdp.h")

F32, BF16, F32X6 = 0, 1, 2          # F32X6: NFDP_F32X6, a `compute` value of the forward convs (see set_f32_products)
_COMPUTE_NAMES = {"f32": F32, "fp32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16}
_default_compute = os.environ.get("NFDP_COMPUTE", "f32")


class NfdpError(RuntimeError):
    pass


def set_default_compute(name: str) -> None:
    """'f32' (exact-fp32 MFMA, parity mode) or 'bf16' (bf16 MFMA, fp32 accumulate; speed mode)."""
    global _default_compute
    if name not in _COMPUTE_NAMES:
        raise ValueError(f"unknown compute mode {name!r}")
    _default_compute = name


def get_default_compute() -> str:
    return _default_compute


def compute_enum(name: Optional[str]) -> int:
    name = name or _default_compute
    if name not in _COMPUTE_NAMES:
        raise ValueError(f"unknown compute mode {name!r}")
    return _COMPUTE_NAMES[name]


# ----------------------------------------------------------------------------- header-driven binding
_CTYPE = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "nfdp_stream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER_PATH) -> Dict[str, tuple]:
    """{name: (restype, [argtypes])} for every `nfdp_*` prototype in include/nfdp.h."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(nfdp_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        restype = ctypes.c_char_p if "char" in ret else _CTYPE[ret.replace("const", "").strip()]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPE[a.replace("const", "").split()[0]])
        protos[name] = (restype, argtypes)
    return protos


_lib = None


def lib():
    """The loaded shared library; raises NfdpError if it has not been built (python -m nfdp_hip.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NfdpError(f"{LIB_PATH} not found: build it with __graft_entry__.build() / nfdp_hip/build.py. "
                            "There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(l, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = l
    return _lib


def _check(rc: int) -> None:
    if rc != 0:
        raise NfdpError(f"nfdp error {rc}: {lib().nfdp_last_error().decode()}")


# ---- optional launch tracing (bench.py / profiling only): HIP events on the launch stream around every kernel
_trace = None


# Kernel-layout copies of weights (conv packs, fused-kernel images) are cached per (storage, tensor version).  A HIP
# kernel that updates parameters through an alias (nfdp_hip.optim.FusedAdam on its flat buffer) is invisible to torch's
# version counters, so it bumps this epoch instead and every cache includes it in its key.
_param_epoch = 0


# How compute = F32 forward convs form their products: a HOST-side default of this binding, handed to the library per call as the
# `compute` argument (NFDP_F32 | NFDP_F32X6) - the library itself keeps no mode.
_F32_PRODUCTS = "bf16x6"


def set_f32_products(mode: str) -> None:
    """'exact' (v_mfma_f32_16x16x4_f32) | 'bf16x6' (default: three-way bf16 split of both operands, six bf16 MFMAs, fp32-chain
    accuracy) for the fp32 forward convs (compute = NFDP_F32X6 in the calls)."""
    global _F32_PRODUCTS
    if mode not in ("exact", "bf16x6"):
        raise ValueError(f"set_f32_products: {mode!r} (exact | bf16x6)")
    _F32_PRODUCTS = mode


def get_f32_products() -> str:
    return _F32_PRODUCTS


def _f32_compute(compute: int) -> int:
    return F32X6 if compute == F32 and _F32_PRODUCTS == "bf16x6" else compute


def param_epoch() -> int:
    return _param_epoch


def bump_param_epoch() -> None:
    global _param_epoch
    _param_epoch += 1


def start_trace() -> None:
    global _trace
    _trace = []


def stop_trace():
    """-> list of dicts {label, ms, flops, bytes}; synchronises the device."""
    global _trace
    t, _trace = _trace, None
    torch.cuda.synchronize()
    return [{"label": lb, "ms": e0.elapsed_time(e1), "flops": fl, "bytes": by} for lb, e0, e1, fl, by in (t or [])]


def _call(name: str, *args, label: Optional[str] = None, flops: float = 0.0, nbytes: float = 0.0) -> None:
    fn = getattr(lib(), name)
    if _trace is None:
        _check(fn(*args))
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _check(fn(*args))
    e1.record()
    _trace.append((label or name, e0, e1, flops, nbytes))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_empty_scratch: Dict[str, torch.Tensor] = {}


def _same_device(t) -> None:
    """Every launch goes to the CURRENT device's current stream (_stream()): a tensor of another GPU would be read / written by
    a kernel queued on the wrong device, with no stream ordering against the work that produced it.  One process per GPU:
    torch.cuda.set_device(LOCAL_RANK) (normalizing_flow.base.default_device() does it when LOCAL_RANK is set), or
    `with torch.cuda.device(t.device)` around the call.  Checked for every pointer handed to the C ABI."""
    cur = torch.cuda.current_device()
    if t.device.index is not None and t.device.index != cur:
        raise NfdpError(f"tensor is on cuda:{t.device.index} but the current device is cuda:{cur}: kernels are queued on the current "
                        f"device's stream - call torch.cuda.set_device({t.device.index}) (one process per GPU) first")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address for the C ABI.  Empty tensors (an empty shard of a ragged data-parallel split) have a null data pointer,
    which the ABI rejects as a missing argument; they get the address of a small per-device scratch instead - every entry
    point returns before touching memory when a size is 0."""
    if t is None:
        return None
    if t.is_cuda:
        _same_device(t)
    if t.numel() == 0 and t.is_cuda:
        key = str(t.device)
        if key not in _empty_scratch:
            _empty_scratch[key] = torch.zeros(64, device=t.device, dtype=torch.uint8)
        return _empty_scratch[key].data_ptr()
    return t.data_ptr()


def _need(t: torch.Tensor, dtype, what: str) -> None:
    if not t.is_cuda:
        raise NfdpError(f"{what}: tensor is on {t.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if t.dtype != dtype:
        raise NfdpError(f"{what}: expected {dtype}, got {t.dtype}")


def _dense(t: torch.Tensor, what: str) -> None:
    if not t.is_contiguous():
        raise NfdpError(f"{what}: tensor must be contiguous")


# ----------------------------------------------------------------------------- NHWC plumbing
def is_nhwc_view(x: torch.Tensor) -> bool:
    """True if logical-NCHW `x` is physically NHWC-dense (channels_last) so permute(0,2,3,1) is contiguous."""
    return x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Logical NCHW fp32 tensor -> physical [B,H,W,C] contiguous tensor (zero-copy when already channels_last)."""
    _need(x, torch.float32, "to_nhwc")
    if x.numel() == 0:                                   # empty shard: strides of empty tensors carry no layout information
        b, c, h, w = x.shape
        return torch.empty((b, h, w, c), device=x.device, dtype=torch.float32)
    if is_nhwc_view(x):
        return x.permute(0, 2, 3, 1)
    if torch.is_grad_enabled() and x.requires_grad:
        # a tensor that carries a gradient (latents of an unfrozen flow, a normalised image) must stay inside autograd: the HIP copy
        # below is invisible to it and would silently cut the graph.  permute + contiguous are tracked views / copies (plumbing).
        return x.permute(0, 2, 3, 1).contiguous()
    x = x.contiguous()
    b, c, h, w = x.shape
    y = torch.empty((b, h, w, c), device=x.device, dtype=torch.float32)
    _call("nfdp_nchw_to_nhwc_f32", _ptr(x), _ptr(y), b, c, h, w, _stream())
    return y


def as_nchw(xh: torch.Tensor) -> torch.Tensor:
    """[B,H,W,C] physical -> logical NCHW view (channels_last strides, zero-copy)."""
    return xh.permute(0, 3, 1, 2)


def to_nchw_contiguous(xh: torch.Tensor) -> torch.Tensor:
    _need(xh, torch.float32, "to_nchw_contiguous")
    _dense(xh, "to_nchw_contiguous")
    b, h, w, c = xh.shape
    y = torch.empty((b, c, h, w), device=xh.device, dtype=torch.float32)
    _call("nfdp_nhwc_to_nchw_f32", _ptr(xh), _ptr(y), b, c, h, w, _stream())
    return y


def _ld(xh: torch.Tensor, what: str) -> int:
    """Pixel stride (elements) of a [B,H,W,C] tensor that may be a channel slice of a wider dense tensor."""
    b, h, w, c = xh.shape
    if xh.numel() == 0:
        return c
    sb, sh, sw, sc = xh.stride()
    if c > 1 and sc != 1:
        raise NfdpError(f"{what}: channel stride must be 1")
    ld = sw if w > 1 else (sh if h > 1 else (sb if b > 1 else c))
    ok = (w == 1 or sw == ld) and (h == 1 or sh == ld * w) and (b == 1 or sb == ld * w * h)
    if not ok or ld < c:
        raise NfdpError(f"{what}: tensor is not a pixel-strided NHWC view (shape {tuple(xh.shape)}, strides {xh.stride()})")
    return ld


def squeeze(xh: torch.Tensor) -> torch.Tensor:
    _need(xh, torch.float32, "squeeze")
    b, h, w, c = xh.shape
    if h % 2 or w % 2:
        raise ValueError("The input dimension is not divisible by 2!")
    y = torch.empty((b, h // 2, w // 2, 4 * c), device=xh.device, dtype=torch.float32)
    _call("nfdp_squeeze_nhwc_f32", _ptr(xh), _ld(xh, "squeeze"), _ptr(y), b, c, h, w, _stream())
    return y


def unsqueeze(yh: torch.Tensor) -> torch.Tensor:
    _need(yh, torch.float32, "unsqueeze")
    _dense(yh, "unsqueeze")
    b, h, w, c4 = yh.shape
    x = torch.empty((b, 2 * h, 2 * w, c4 // 4), device=yh.device, dtype=torch.float32)
    _call("nfdp_unsqueeze_nhwc_f32", _ptr(yh), _ptr(x), b, c4, h, w, _stream())
    return x


def copy_channels(src: torch.Tensor, dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dst[..., :] = src[..., :] for two [B,H,W,n] pixel-strided views (dst defaults to a new dense tensor)."""
    _need(src, torch.float32, "copy_channels")
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=torch.float32)
    _need(dst, torch.float32, "copy_channels")
    b, h, w, n = src.shape
    assert dst.shape == src.shape
    _call("nfdp_copy_channels_f32", _ptr(src), _ld(src, "copy_channels"), _ptr(dst), _ld(dst, "copy_channels"),
                                        b * h * w, n, _stream())
    return dst


# ----------------------------------------------------------------------------- ActNorm / InvConv
def actnorm_init(xh: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor) -> None:
    """Writes data-dependent scale/bias (flat [C] views of the parameters) from the batch."""
    b, h, w, c = xh.shape
    ws = torch.empty(lib().nfdp_actnorm_init_ws_bytes(c), device=xh.device, dtype=torch.uint8)
    _call("nfdp_actnorm_init_f32", _ptr(xh), _ld(xh, "actnorm_init"), b * h * w, c, _ptr(scale), _ptr(bias), _ptr(ws),
                                       _stream())


def actnorm_fwd(xh, scale, bias, out=None):
    _need(xh, torch.float32, "actnorm_fwd"); _dense(xh, "actnorm_fwd")
    out = torch.empty_like(xh) if out is None else out
    c = xh.shape[-1]
    _call("nfdp_actnorm_fwd_f32", _ptr(xh), _ptr(out), xh.numel() // c, c, _ptr(scale), _ptr(bias), _stream(),
          label=f"actnorm_kernel<fwd> C{c}", nbytes=8.0 * xh.numel())
    return out


def actnorm_inv(yh, scale, bias, out=None):
    _need(yh, torch.float32, "actnorm_inv"); _dense(yh, "actnorm_inv")
    out = torch.empty_like(yh) if out is None else out
    c = yh.shape[-1]
    _call("nfdp_actnorm_inv_f32", _ptr(yh), _ptr(out), yh.numel() // c, c, _ptr(scale), _ptr(bias), _stream())
    return out


def invconv_prepare(w: torch.Tensor, want_inverse: bool):
    """w: [n,C,C] (or [C,C]) fp32 -> (logabsdet [n] fp32, winv [n,C,C] fp32 or None); one launch for all n."""
    _need(w, torch.float32, "invconv_prepare"); _dense(w, "invconv_prepare")
    c = w.shape[-1]
    n = w.numel() // (c * c)
    lad = torch.empty(n, device=w.device, dtype=torch.float32)
    winv = torch.empty((n, c, c), device=w.device, dtype=torch.float32) if want_inverse else None
    ws = torch.empty(lib().nfdp_invconv_prepare_ws_bytes(n, c), device=w.device, dtype=torch.uint8)
    _call("nfdp_invconv_prepare_f32", _ptr(w), n, c, _ptr(lad), _ptr(winv), _ptr(ws), _stream())
    return lad, winv


def invconv_logdets(ws) -> torch.Tensor:
    """log|det W| (fp64 on the device, fp32 result [n]) of a list of square matrices of any sizes <= 96, ONE launch."""
    n = len(ws)
    for w in ws:
        _need(w, torch.float32, "invconv_logdets"); _dense(w, "invconv_logdets")
    out = torch.empty(n, device=ws[0].device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    cs = (ctypes.c_int * n)(*[w.shape[-1] for w in ws])
    _call("nfdp_invconv_logdets_f32", ptrs, cs, n, _ptr(out), _stream())
    return out


def invconv_inverses_supported(c: int) -> bool:
    return bool(lib().nfdp_invconv_inverses_supported(c))


def invconv_inverses(groups) -> list:
    """W^-1 of several GROUPS of square matrices in ONE launch (nfdp_invconv_inverses_f32): groups = [[w, w, ...], ...], the matrices
    of a group of one size (a level's 1x1 convs); -> one [K, C, C] tensor per group (fp64 Gauss-Jordan on the device, as
    invconv_prepare: same results)."""
    flat, cs, outs, optrs = [], [], [], []
    for g in groups:
        c = g[0].shape[-1]
        out = torch.empty((len(g), c, c), device=g[0].device, dtype=torch.float32)
        outs.append(out)
        for i, w in enumerate(g):
            _need(w, torch.float32, "invconv_inverses"); _dense(w, "invconv_inverses")
            if w.shape[-1] != c or w.shape[-2] != c:
                raise ValueError("invconv_inverses: a group holds matrices of one size")
            flat.append(w.data_ptr()); cs.append(c); optrs.append(out.data_ptr() + i * c * c * 4)
    n = len(flat)
    if n == 0:
        return outs
    lad = torch.empty(n, device=groups[0][0].device, dtype=torch.float32)
    _call("nfdp_invconv_inverses_f32", (ctypes.c_void_p * n)(*flat), (ctypes.c_int * n)(*cs), n, _ptr(lad), (ctypes.c_void_p * n)(*optrs),
          _stream())
    return outs


def invconv_apply(xh, w, out=None):
    _need(xh, torch.float32, "invconv_apply"); _dense(xh, "invconv_apply")
    out = torch.empty_like(xh) if out is None else out
    c = xh.shape[-1]
    _call("nfdp_invconv_apply_f32", _ptr(xh), _ptr(out), xh.numel() // c, c, _ptr(w), _stream(),
          label=f"pixel_linear_kernel<{c},matvec>", flops=2.0 * xh.numel() * c, nbytes=8.0 * xh.numel())
    return out


_WIDE_C = 96      # from here on the per-pixel CxC product runs on the exact-fp32 MFMA conv kernel (a lane of pixel_linear_kernel would
                  # walk C/4 x C serial FMAs: 142 us for the 512 pixels of the C = 192 level of a 5-level model)


def _mfma_1x1(xh, w, out):
    c = xh.shape[-1]
    y = conv2d(xh, conv_pack(w.detach().reshape(c, c, 1, 1), F32), c, 1, compute=F32)
    out.copy_(y)
    return out


def actnorm_invconv_fwd(xh, scale, bias, w, out=None):
    _need(xh, torch.float32, "actnorm_invconv_fwd"); _dense(xh, "actnorm_invconv_fwd")
    out = torch.empty_like(xh) if out is None else out
    c = xh.shape[-1]
    if c >= _WIDE_C and c % 8 == 0 and c not in _FLOW_HEAD_C:
        return _mfma_1x1(actnorm_fwd(xh, scale, bias), w, out)
    _call("nfdp_actnorm_invconv_fwd_f32", _ptr(xh), _ptr(out), xh.numel() // c, c, _ptr(scale), _ptr(bias), _ptr(w),
          _stream(), label=f"pixel_linear_kernel<{c},pre_affine>", flops=2.0 * xh.numel() * c, nbytes=8.0 * xh.numel())
    return out


_FLOW_HEAD_C = (4, 8, 12, 16, 24, 32, 48, 64, 96, 192)      # the templated instances of pixel_linear.hip (64 / 96 / 192: fp32 MFMA)


def flow_head_fwd(xh, scale, bias, w, logabsdet, out=None):
    """ActNorm -> 1x1 conv in one pass (in place if out is xh) and the flow's two log-det constants as a device
    fp32 [2] tensor (consumed by coupling_fwd / coupling_fused).  Falls back to two launches for untemplated C."""
    _need(xh, torch.float32, "flow_head_fwd"); _dense(xh, "flow_head_fwd")
    out = torch.empty_like(xh) if out is None else out
    b, h, wd, c = xh.shape
    if c not in _FLOW_HEAD_C:
        if logabsdet is None:
            logabsdet = torch.zeros(1, device=xh.device, dtype=torch.float32)
        cst = flow_ldj_consts(scale, logabsdet, h * wd)
        actnorm_invconv_fwd(xh, scale, bias, w, out=out)
        return out, cst
    cst = torch.empty(2, device=xh.device, dtype=torch.float32)
    _call("nfdp_flow_head_fwd_f32", _ptr(xh), _ptr(out), xh.numel() // c, c, _ptr(scale), _ptr(bias), _ptr(w),
          _ptr(logabsdet), float(h * wd), _ptr(cst), _stream(),
          label=f"pixel_linear_kernel<{c},pre_affine>", flops=2.0 * xh.numel() * c, nbytes=8.0 * xh.numel())
    return out, cst


def invconv_actnorm_inv(yh, scale, bias, winv, out=None):
    _need(yh, torch.float32, "invconv_actnorm_inv"); _dense(yh, "invconv_actnorm_inv")
    out = torch.empty_like(yh) if out is None else out
    c = yh.shape[-1]
    if c >= _WIDE_C and c % 8 == 0 and c not in _FLOW_HEAD_C:
        _mfma_1x1(yh, winv, out)
        return actnorm_inv(out, scale, bias, out=out)
    _call("nfdp_invconv_actnorm_inv_f32", _ptr(yh), _ptr(out), yh.numel() // c, c, _ptr(scale), _ptr(bias), _ptr(winv),
          _stream(), label=f"pixel_linear_kernel<{c},post_affine>", flops=2.0 * yh.numel() * c, nbytes=8.0 * yh.numel())
    return out


# ----------------------------------------------------------------------------- convolution
def conv_pack(w_oihw: torch.Tensor, compute: int) -> torch.Tensor:
    _need(w_oihw, torch.float32, "conv_pack"); _dense(w_oihw, "conv_pack")
    co, ci, kh, kw = w_oihw.shape
    nbytes = lib().nfdp_conv_packed_bytes(co, ci, kh, kw, compute)
    packed = torch.empty(nbytes, device=w_oihw.device, dtype=torch.uint8)
    _call("nfdp_conv_pack_weights", _ptr(w_oihw), _ptr(packed), co, ci, kh, kw, compute, _stream())
    return packed


_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def use_im2col_conv(cin: int, cout: int, k: int, compute: int) -> bool:
    """The narrow-input 3x3 convs of the training path (C/2 -> 512): explicit im2col (a matrix 9C wide, a tenth of the
    512-channel tensors it meets) + plain K-contiguous GEMMs for the conv and its weight gradient."""
    return compute == BF16 and k == 3 and 9 * cin <= 512 and cout > 64 and cout % 4 == 0


def im2col_kpad(cin: int) -> int:
    return -(-9 * cin // 64) * 64


def im2col3x3(xh: torch.Tensor, pad_batch: int = 0) -> torch.Tensor:
    """xh [B,H,W,C] fp32 pixel-strided view -> [B,H,W,Kpad] bf16, columns (tap, c), zero padded (nfdp_im2col3x3_bf16).  Kpad is
    the row pitch of conv_pack's layout for the 3x3 weight, so that packed weight is also the 1x1 weight for these rows.
    pad_batch > B: the result has pad_batch images, the extra ones all zero (operands of GEMMs over whole workgroups' pixels)."""
    _need(xh, torch.float32, "im2col3x3")
    b, h, w, c = xh.shape
    kpad = im2col_kpad(c)
    if pad_batch > b:
        out = torch.empty((pad_batch, h, w, kpad), device=xh.device, dtype=torch.bfloat16)
        out[b:].zero_()
    else:
        out = torch.empty((b, h, w, kpad), device=xh.device, dtype=torch.bfloat16)
    _call("nfdp_im2col3x3_bf16", _ptr(xh), _ld(xh, "im2col3x3"), _ptr(out), b, h, w, c, kpad, _stream(), label="im2col3x3",
          nbytes=float(b * h * w) * (c * 4 * 9 + kpad * 2))
    return out


def conv2d(xh: torch.Tensor, packed: torch.Tensor, cout: int, k: int, *, add=None, mul=None, mul_exp: float = 0.0,
           relu: bool = False, compute: int = F32, out_dtype: int = F32, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """xh: [B,H,W,Cin] pixel-strided view (fp32 or bf16) -> [B,H,W,cout] (fp32 or bf16)."""
    if not xh.is_cuda:
        raise NfdpError("conv2d: the HIP path needs a GPU tensor (no CPU fallback)")
    b, h, w, cin = xh.shape
    x_dt = BF16 if xh.dtype == torch.bfloat16 else F32
    if out is None:
        out = torch.empty((b, h, w, cout), device=xh.device, dtype=_TORCH_DT[out_dtype])
    lbl = None
    if _trace is not None:
        bn = 16 if cout <= 16 else 32 if cout <= 32 else 48 if cout <= 48 else 64 if cout <= 64 else 128
        vec = cin % 4 == 0 and _ld(xh, "conv2d") % 4 == 0
        lbl = (f"conv_kernel<{'bf16' if compute == BF16 else 'f32'},x{'bf16' if x_dt == BF16 else 'f32'},"
               f"{'vec' if vec else 'scalar'},{256 if bn <= 48 else 128}x{bn}> k{k} {cin}->{cout} hw{h}")
        if compute == F32 and cin % 8 == 0 and cout >= 64 and cout % 4 == 0 and cin * k * k >= 256 and \
                -(-b * h * w // 64) * -(-cout // 64) <= 1024:
            lbl = f"conv_splitk_f32 k{k} {cin}->{cout} hw{h}"
        if compute == BF16 and x_dt == BF16 and k == 1 and cin % 64 == 0 and cout > 64 and cout % 4 == 0 and b * h * w >= 512:
            lbl = f"pw_gemm_bf16 k1 {cin}->{cout} hw{h}"
    _call("nfdp_conv2d_nhwc", _ptr(xh), x_dt, _ld(xh, "conv2d"), _ptr(packed), _ptr(out), out_dtype, _ld(out, "conv2d"),
          b, h, w, cin, cout, k, k, _ptr(add), _ptr(mul), float(mul_exp), int(relu), _f32_compute(compute), _stream(),
          label=lbl, flops=2.0 * b * h * w * cin * k * k * cout,
          nbytes=float(b * h * w) * (cin * xh.element_size() + cout * out.element_size()))
    return out


# ----------------------------------------------------------------------------- coupling / priors / NLL
def coupling_fwd(xh, hh, ldj: Optional[torch.Tensor], ldj_const: Optional[torch.Tensor]):
    _dense(xh, "coupling_fwd"); _dense(hh, "coupling_fwd")
    b, h, w, c = xh.shape
    if ldj is not None:
        _need(ldj, torch.float64, "coupling_fwd ldj")
    nbytes = lib().nfdp_coupling_chunked_ws_bytes(b, h * w, c)
    if nbytes:                                           # large image, small batch: several workgroups per sample
        ws = torch.empty(nbytes, device=xh.device, dtype=torch.uint8)
        _call("nfdp_coupling_fwd_chunked_f32", _ptr(xh), _ptr(hh), b, h * w, c, _ptr(ldj), _ptr(ldj_const), _ptr(ws), _stream(),
              label="coupling_kernel<fwd,chunked>", nbytes=4.0 * xh.numel() * 2)
        return
    _call("nfdp_coupling_fwd_f32", _ptr(xh), _ptr(hh), b, h * w, c, _ptr(ldj), _ptr(ldj_const), _stream(),
          label="coupling_kernel<fwd>", nbytes=4.0 * xh.numel() * 2)


def tap_coupling(xh, t, add, mul, mul_exp: float, ldj: Optional[torch.Tensor], ldj_const: Optional[torch.Tensor], inverse: bool) -> None:
    """Tap shift-sum of the ZeroConv2d's tap contributions `t` [B,H,W,9*C] + affine coupling on xh [B,H,W,C] in place, one pass."""
    _dense(xh, "tap_coupling"); _dense(t, "tap_coupling")
    _need(xh, torch.float32, "tap_coupling"); _need(t, torch.float32, "tap_coupling")
    b, h, w, c = xh.shape
    if t.shape[-1] != 9 * c:
        raise ValueError(f"tap_coupling: {t.shape[-1]} tap contributions for {c} channels")
    ws = None
    if not inverse and ldj is not None:
        _need(ldj, torch.float64, "tap_coupling ldj")
        nbytes = lib().nfdp_tap_coupling_ws_bytes(b, h * w, c)
        if nbytes:
            ws = torch.empty(nbytes, device=xh.device, dtype=torch.uint8)
    _call("nfdp_tap_coupling_f32", _ptr(xh), _ptr(t), b, h, w, c, _ptr(add), _ptr(mul), float(mul_exp), _ptr(None if inverse else ldj),
          _ptr(None if inverse else ldj_const), _ptr(ws), int(inverse), _stream(),
          label=f"tap_coupling<{'inv' if inverse else 'fwd'}>", nbytes=4.0 * (t.numel() + xh.numel()))


TAP_HEAD_FUSION = True      # tiled levels: the adjoining flow head inside the tap + coupling pass (False: two launches per flow)


def tap_coupling_head_supported(c: int) -> bool:
    return TAP_HEAD_FUSION and bool(lib().nfdp_tap_coupling_head_supported(c))


def tap_coupling_head(xh, t, add, mul, mul_exp: float, ldj: Optional[torch.Tensor], ldj_const: Optional[torch.Tensor], inverse: bool,
                      head_scale, head_bias, head_w, head_logabsdet=None, want_const: bool = False, defer: Optional[list] = None):
    """tap_coupling with the adjoining flow head in the same pass (nfdp_tap_coupling_head_f32).  Forward: (head_scale, head_bias,
    head_w) are the NEXT flow's ActNorm / 1x1-conv parameters; returns that flow's [2] log-det constants when want_const.  Inverse: THIS
    flow's ActNorm parameters and W^-1.  defer (a list, forward with ldj): the launch leaves its log-det chunk sums in a workspace
    appended to the list as (ws, ldj_const, nch) instead of adding them to ldj; ldj_finish_level(ldj, defer) adds a level's launches up
    in one launch (bit-identical to the per-launch finishes)."""
    _dense(xh, "tap_coupling_head"); _dense(t, "tap_coupling_head")
    _need(xh, torch.float32, "tap_coupling_head"); _need(t, torch.float32, "tap_coupling_head")
    b, h, w, c = xh.shape
    if t.shape[-1] != 9 * c:
        raise ValueError(f"tap_coupling_head: {t.shape[-1]} tap contributions for {c} channels")
    for q in (head_scale, head_bias, head_w):
        _need(q, torch.float32, "tap_coupling_head"); _dense(q, "tap_coupling_head")
    if head_scale.numel() != c or head_bias.numel() != c or head_w.numel() != c * c:
        raise ValueError("tap_coupling_head: head parameter shapes")
    ws = None
    deferring = False
    if not inverse and ldj is not None:
        _need(ldj, torch.float64, "tap_coupling_head ldj")
        nch = int(lib().nfdp_tap_coupling_head_chunks(b, h * w, c))
        deferring = defer is not None and nch > 1        # (one chunk: the launch adds its sum to ldj itself, nothing to save)
        if nch > 1:
            ws = torch.empty(b * nch, device=xh.device, dtype=torch.float64)
        if deferring:
            defer.append((ws, ldj_const, nch))
    cst = torch.empty(2, device=xh.device, dtype=torch.float32) if (want_const and not inverse) else None
    _call("nfdp_tap_coupling_head_f32", _ptr(xh), _ptr(t), b, h, w, c, _ptr(add), _ptr(mul), float(mul_exp),
          _ptr(None if inverse else ldj), _ptr(None if inverse else ldj_const), _ptr(ws), int(inverse), _ptr(head_scale), _ptr(head_bias),
          _ptr(head_w), _ptr(None if inverse else head_logabsdet), _ptr(cst), int(deferring), _stream(),
          label=f"tap_coupling_head<{'inv' if inverse else 'fwd'},{c}>", nbytes=4.0 * (t.numel() + 2 * xh.numel()),
          flops=2.0 * xh.numel() * c)
    return cst


def ldj_finish_level(ldj: torch.Tensor, deferred: list) -> None:
    """Adds the deferred log-det sums of a level's tap_coupling_head launches to ldj (one launch; nfdp_ldj_finish_level_f64)."""
    if not deferred:
        return
    _need(ldj, torch.float64, "ldj_finish_level")
    n, nch = len(deferred), deferred[0][2]
    if any(d[2] != nch for d in deferred):
        raise ValueError("ldj_finish_level: launches of different geometries")
    parts = (ctypes.c_void_p * n)(*[_ptr(d[0]) for d in deferred])
    csts = (ctypes.c_void_p * n)(*[None if d[1] is None else _ptr(d[1]) for d in deferred])
    _call("nfdp_ldj_finish_level_f64", _ptr(ldj), ldj.numel(), parts, csts, n, nch, _stream())
    deferred.clear()


def coupling_inv(yh, hh):
    _dense(yh, "coupling_inv"); _dense(hh, "coupling_inv")
    b, h, w, c = yh.shape
    _call("nfdp_coupling_inv_f32", _ptr(yh), _ptr(hh), b, h * w, c, _stream())


def coupling_fused_supported(h: int, w: int, c: int, n_features: int) -> bool:
    return bool(lib().nfdp_coupling_fused_supported(h, w, c, n_features))


def coupling_fused_infer_supported(h: int, w: int, c: int, n_features: int) -> bool:
    """Geometries the fused flow-step kernels run in inference (the training kernels' set + the wide levels of 5-level models)."""
    return bool(lib().nfdp_coupling_fused_infer_supported(h, w, c, n_features))


def coupling_fused_has_head(h: int, w: int, c: int) -> bool:
    """Does the fused launch of this geometry include the flow head (ActNorm + 1x1 conv)?  False: coupling layer only."""
    return bool(lib().nfdp_coupling_fused_has_head(h, w, c))


def coupling_fused_image_kind(h: int, w: int, c: int) -> int:
    """Which inference image the fused kernels of this geometry read: 1 (fold=True) or 2 (fold=2: 16x16x32 fragments); 0 = unsupported."""
    return int(lib().nfdp_coupling_fused_image_kind(h, w, c))


def coupling_fused_pack(w1, s1, b1, w2, s2, b2, w3, b3, l3, fold=True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fold=True (1): the inference image (hidden ActNorm scales folded into the next GEMM's weights); fold=2: the same in the fragment
    layout of the 16x16 levels' kernel (coupling_fused_image_kind); fold=False: the image of the training forward (level_fused_train).
    `out`: re-use an existing image buffer."""
    c = w3.shape[0]
    nbytes = lib().nfdp_coupling_fused_packed_bytes(c)
    packed = out if out is not None and out.numel() == nbytes else torch.empty(nbytes, device=w1.device, dtype=torch.uint8)
    for t in (w1, s1, b1, w2, s2, b2, w3, b3, l3):
        _need(t, torch.float32, "coupling_fused_pack"); _dense(t, "coupling_fused_pack")
    _call("nfdp_coupling_fused_pack_ex", _ptr(w1), _ptr(s1), _ptr(b1), _ptr(w2), _ptr(s2), _ptr(b2), _ptr(w3), _ptr(b3),
          _ptr(l3), _ptr(packed), c, int(fold), _stream())
    return packed


def _pack_batch(entry: str, per_net: int, nets, outs, c: int, nbytes: int, *extra) -> list:
    n = len(nets)
    if n == 0:
        return []
    dev = nets[0][0].device
    if outs is None:
        outs = [None] * n
    outs = [o if o is not None and o.numel() == nbytes else torch.empty(nbytes, device=dev, dtype=torch.uint8) for o in outs]
    flat = []
    for ts in nets:
        if len(ts) != per_net:
            raise ValueError(f"{entry}: {per_net} tensors per network")
        for t in ts:
            _need(t, torch.float32, entry); _dense(t, entry)
            flat.append(_ptr(t))
    for i in range(0, n, 32):                                        # (the launch takes up to 32 networks)
        m = min(32, n - i)
        _call(entry, (ctypes.c_void_p * (per_net * m))(*flat[per_net * i: per_net * (i + m)]),
              (ctypes.c_void_p * m)(*[_ptr(o) for o in outs[i: i + m]]), m, c, *extra, _stream())
    return outs


def coupling_fused_pack_batch(nets, fold: bool = True, outs=None) -> list:
    """coupling_fused_pack for several networks of one geometry in one launch: nets = [(w1, s1, b1, w2, s2, b2, w3, b3, l3), ...];
    outs: optional list of image buffers to re-use (None entries are allocated).  -> the images."""
    if not nets:
        return []
    c = nets[0][6].shape[0]
    return _pack_batch("nfdp_coupling_fused_pack_batch", 9, nets, outs, c, lib().nfdp_coupling_fused_packed_bytes(c), int(fold))


def coupling_bwd_fused_pack_batch(nets, outs=None) -> list:
    """coupling_bwd_fused_pack for several networks in one launch: nets = [(w1, s1, w2, s2, w3), ...]."""
    if not nets:
        return []
    c = nets[0][4].shape[0]
    nbytes = lib().nfdp_coupling_bwd_fused_packed_bytes(c)
    if nbytes == 0:
        raise NfdpError(f"coupling_bwd_fused_pack_batch: unsupported C={c}")
    return _pack_batch("nfdp_coupling_bwd_fused_pack_batch", 5, nets, outs, c, nbytes)


WIDE_LEVELS_SPLIT = True     # the 8x8x96 / 4x4x192 levels: a pixel tile's GEMM rows on four workgroups, three launches per flow


def set_wide_levels_split(on: bool) -> None:
    """False: the one-workgroup-per-tile kernels (coupling_fused_wide_kernel) - A/B runs and the comparison test."""
    global WIDE_LEVELS_SPLIT
    WIDE_LEVELS_SPLIT = bool(on)


def _split_ws_bytes(b: int, h: int, w: int, c: int) -> int:
    return int(lib().nfdp_level_split_ws_bytes(b, h, w, c)) if WIDE_LEVELS_SPLIT else 0


def level_fused_takes_heads(h: int, w: int, c: int) -> bool:
    """Can level_fused run this geometry's flows WITH their flow heads?  The one-workgroup kernels: coupling_fused_has_head (not at
    4x4x192); the split launches of the wide levels take the head at both widths."""
    return bool(coupling_fused_has_head(h, w, c)) or _split_ws_bytes(1, h, w, c) > 0


def coupling_fused(xh: torch.Tensor, packed: torch.Tensor, ldj: Optional[torch.Tensor], ldj_const: Optional[torch.Tensor],
                   inverse: bool) -> None:
    """In place on xh [B,H,W,C]: the whole coupling layer (network + affine + log-det) in one launch (bf16 MFMA)."""
    _need(xh, torch.float32, "coupling_fused"); _dense(xh, "coupling_fused")
    b, h, w, c = xh.shape
    if ldj is not None:
        _need(ldj, torch.float64, "coupling_fused ldj")
    ch = c // 2
    flops = 2.0 * b * h * w * (9 * ch * 512 + 512 * 512 + 9 * 512 * c)
    wsb = _split_ws_bytes(b, h, w, c)
    if wsb:                                              # wide level: a pixel tile on four workgroups, three launches (fused_split.inc)
        ws = torch.empty(wsb, device=xh.device, dtype=torch.uint8)
        _call("nfdp_coupling_fused_ws_bf16", _ptr(xh), _ptr(packed), b, h, w, c, _ptr(ldj), _ptr(ldj_const), int(inverse), _ptr(ws),
              wsb, _stream(), label=f"coupling_split<{c},{w}> {'inv' if inverse else 'fwd'}", flops=flops, nbytes=8.0 * xh.numel())
        return
    _call("nfdp_coupling_fused_bf16", _ptr(xh), _ptr(packed), b, h, w, c, _ptr(ldj), _ptr(ldj_const), int(inverse),
          _stream(), label=f"coupling_fused_kernel<{c},{w}> {'inv' if inverse else 'fwd'}", flops=flops,
          nbytes=8.0 * xh.numel())


def stepflow_fused(xh: torch.Tensor, packed: torch.Tensor, scale, bias, w, ldj: Optional[torch.Tensor],
                   logdet_term: Optional[torch.Tensor], inverse: bool) -> None:
    """In place on xh [B,H,W,C]: ActNorm -> 1x1 conv -> coupling layer (or the exact reverse) in ONE launch."""
    _need(xh, torch.float32, "stepflow_fused"); _dense(xh, "stepflow_fused")
    b, h, wd, c = xh.shape
    if ldj is not None:
        _need(ldj, torch.float64, "stepflow_fused ldj")
    ch = c // 2
    flops = 2.0 * b * h * wd * (9 * ch * 512 + 512 * 512 + 9 * 512 * c + c * c)
    _call("nfdp_stepflow_fused_bf16", _ptr(xh), _ptr(packed), b, h, wd, c, _ptr(scale), _ptr(bias), _ptr(w), _ptr(ldj),
          _ptr(logdet_term), int(inverse), _stream(),
          label=f"coupling_fused_kernel<{c},{wd}>+head {'inv' if inverse else 'fwd'}", flops=flops, nbytes=8.0 * xh.numel())


def level_fused(xh: torch.Tensor, packed, scales, biases, ws, ldj: Optional[torch.Tensor], inverse: bool) -> None:
    """In place on xh [B,H,W,C]: len(packed) consecutive flow steps (forward order; `ws` holds W, or W^-1 when inverse) in
    ONE persistent launch - the flow-state tile stays in LDS across the flows."""
    _need(xh, torch.float32, "level_fused"); _dense(xh, "level_fused")
    b, h, wd, c = xh.shape
    n = len(packed)
    if not (len(scales) == len(biases) == len(ws) == n):
        raise ValueError("level_fused: pointer tables of different lengths")
    if ldj is not None:
        _need(ldj, torch.float64, "level_fused ldj")
    tab = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    ch = c // 2
    flops = n * 2.0 * b * h * wd * (9 * ch * 512 + 512 * 512 + 9 * 512 * c + c * c)
    wsb = _split_ws_bytes(b, h, wd, c)
    if wsb:                                              # wide level: a pixel tile on four workgroups, three launches per flow
        wsp = torch.empty(wsb, device=xh.device, dtype=torch.uint8)
        _call("nfdp_level_fused_ws_bf16", _ptr(xh), b, h, wd, c, n, tab(packed), tab(scales), tab(biases), tab(ws), _ptr(ldj),
              int(inverse), _ptr(wsp), wsb, _stream(), label=f"level_split<{c},{wd}> x{n} {'inv' if inverse else 'fwd'}", flops=flops,
              nbytes=8.0 * xh.numel())
        return
    _call("nfdp_level_fused_bf16", _ptr(xh), b, h, wd, c, n, tab(packed), tab(scales), tab(biases), tab(ws), _ptr(ldj),
          int(inverse), _stream(), label=f"level_fused<{c},{wd}> x{n} {'inv' if inverse else 'fwd'}", flops=flops,
          nbytes=8.0 * xh.numel())


TRAIN_RELU_MASKS = True      # the training forward also writes the hidden activations' ReLU masks (1/8 of their bytes) for the reverse network


def level_fused_train(xh: torch.Tensor, packed, scales, biases, ws, ldj: Optional[torch.Tensor], keep_hidden: bool = True,
                      want_masks: bool = False):
    """Training forward of a whole Glow level in ONE persistent launch (nfdp_level_fused_train_bf16): xh [B,H,W,C] is updated in
    place; ldj[b] += the flows' per-sample coupling log-dets only.  -> (xpre, ypost, hout [n,B,H,W,C] fp32; y1, y2
    [n,Bpad,H,W,512] bf16; xcol [n,Bpad,H,W,cols] bf16): per flow the tile before / after the flow head, the zero conv's output, the
    two hidden activations - y1 / y2 in the fused kernels' BLOCKED layout (blocked_to_rows), not [pixel][512] rows - and the im2col
    rows of the coupling network's input.  keep_hidden=False: y1 / y2 / xcol are one flow large and every flow overwrites them (returned as
    None): the forward of a step that recomputes them in its backward pass.
    want_masks: a seventh result - the ReLU masks of y1 / y2 ([n, 2, Bpad*H*W*32] int32; None where the level's kernels have none, or with
    keep_hidden=False): what coupling_bwd_fused reads instead of the activations."""
    _need(xh, torch.float32, "level_fused_train"); _dense(xh, "level_fused_train")
    b, h, wd, c = xh.shape
    n = len(packed)
    if not (len(scales) == len(biases) == len(ws) == n):
        raise ValueError("level_fused_train: pointer tables of different lengths")
    if ldj is not None:
        _need(ldj, torch.float64, "level_fused_train ldj")
    grp = lib().nfdp_level_fused_train_group(h, wd, c)
    if grp <= 0:
        raise NfdpError(f"level_fused_train: unsupported geometry {h}x{wd}x{c}")
    bpad = (b + grp - 1) // grp * grp
    dev = xh.device
    xpre = torch.empty((n, b, h, wd, c), device=dev, dtype=torch.float32)
    ypost = torch.empty((n, b, h, wd, c), device=dev, dtype=torch.float32)
    hout = torch.empty((n, b, h, wd, c), device=dev, dtype=torch.float32)
    nh = n if keep_hidden else 1
    y1 = torch.empty((nh, bpad, h, wd, 512), device=dev, dtype=torch.bfloat16)
    y2 = torch.empty((nh, bpad, h, wd, 512), device=dev, dtype=torch.bfloat16)
    xcol = torch.empty((nh, bpad, h, wd, lib().nfdp_coupling_fused_im2col_cols(c, 0)), device=dev, dtype=torch.bfloat16)
    masks = None
    if want_masks and keep_hidden and TRAIN_RELU_MASKS and lib().nfdp_level_fused_train_has_masks(h, wd, c):
        masks = torch.empty((n, 2, bpad * h * wd * 32), device=dev, dtype=torch.int32)
    tab = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    ch = c // 2
    flops = n * 2.0 * b * h * wd * (9 * ch * 512 + 512 * 512 + 9 * 512 * c + c * c)
    _call("nfdp_level_fused_train_ex_bf16", _ptr(xh), b, h, wd, c, n, tab(packed), tab(scales), tab(biases), tab(ws), _ptr(ldj),
          _ptr(xpre), _ptr(ypost), _ptr(hout), _ptr(y1), _ptr(y2), _ptr(xcol), _ptr(masks), int(keep_hidden), _stream(),
          label=f"level_fused_train<{c},{wd}> x{n}", flops=flops,
          nbytes=8.0 * xh.numel() + 12.0 * n * xh.numel() + 2048.0 * n * bpad * h * wd)
    if not keep_hidden:
        return (xpre, ypost, hout, None, None, None) + ((None,) if want_masks else ())
    return (xpre, ypost, hout, y1, y2, xcol) + ((masks,) if want_masks else ())


def coupling_bwd_fused_supported(h: int, w: int, c: int, n_features: int) -> bool:
    return coupling_fused_supported(h, w, c, n_features) and lib().nfdp_coupling_bwd_fused_packed_bytes(c) > 0


def coupling_bwd_fused_pack(w1, s1, w2, s2, w3, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight image of the reverse coupling network (nfdp_coupling_bwd_fused_pack): w1 [512,C/2,3,3], s1 [512], w2 [512,512,1,1],
    s2 [512], w3 [C,512,3,3]."""
    c = w3.shape[0]
    nbytes = lib().nfdp_coupling_bwd_fused_packed_bytes(c)
    if nbytes == 0:
        raise NfdpError(f"coupling_bwd_fused_pack: unsupported C={c}")
    packed = out if out is not None and out.numel() == nbytes else torch.empty(nbytes, device=w1.device, dtype=torch.uint8)
    for t in (w1, s1, w2, s2, w3):
        _need(t, torch.float32, "coupling_bwd_fused_pack"); _dense(t, "coupling_bwd_fused_pack")
    _call("nfdp_coupling_bwd_fused_pack", _ptr(w1), _ptr(s1), _ptr(w2), _ptr(s2), _ptr(w3), _ptr(packed), c, _stream())
    return packed


def coupling_bwd_fused(dc3: torch.Tensor, dx: torch.Tensor, packed: torch.Tensor, y1: torch.Tensor, y2: torch.Tensor,
                       masks: Optional[torch.Tensor] = None):
    """The coupling network's data-gradient chain in one launch (nfdp_coupling_bwd_fused_bf16).  dc3 [B,H,W,C] fp32 (gradient at the
    zero conv's raw output); y1, y2 [Bpad,H,W,512] bf16 as saved by level_fused_train (BLOCKED layout, Bpad >= B: whole workgroups);
    dx [B,H,W,C] fp32: its first C/2 channels receive += the network's input gradient.  -> (dc1, dc2: shaped and laid out like y1;
    dc3col [Bpad,H,W,cols] bf16: im2col rows of dc3, ordinary rows).
    masks: this flow's [2, Bpad*H*W*32] int32 slice of level_fused_train's mask tensor - read instead of y1 / y2 (same bits out)."""
    _need(dc3, torch.float32, "coupling_bwd_fused"); _dense(dc3, "coupling_bwd_fused")
    _need(dx, torch.float32, "coupling_bwd_fused dx"); _dense(dx, "coupling_bwd_fused dx")
    b, h, w, c = dc3.shape
    grp = lib().nfdp_level_fused_train_group(h, w, c)
    bpad = (b + grp - 1) // grp * grp if grp > 0 else b
    if masks is None or y1 is not None or y2 is not None:
        _need(y1, torch.bfloat16, "coupling_bwd_fused y1"); _dense(y1, "coupling_bwd_fused y1")
        _need(y2, torch.bfloat16, "coupling_bwd_fused y2"); _dense(y2, "coupling_bwd_fused y2")
        if tuple(y1.shape) != (bpad, h, w, 512) or tuple(y2.shape) != (bpad, h, w, 512):
            raise ValueError(f"coupling_bwd_fused: shapes dc3 {tuple(dc3.shape)} y1 {tuple(y1.shape)} y2 {tuple(y2.shape)}")
    if tuple(dx.shape) != (b, h, w, c):
        raise ValueError(f"coupling_bwd_fused: shapes dc3 {tuple(dc3.shape)} dx {tuple(dx.shape)}")
    dc1 = torch.empty((bpad, h, w, 512), device=dc3.device, dtype=torch.bfloat16)
    dc2 = torch.empty((bpad, h, w, 512), device=dc3.device, dtype=torch.bfloat16)
    dc3col = torch.empty((bpad, h, w, lib().nfdp_coupling_fused_im2col_cols(c, 1)), device=dc3.device, dtype=torch.bfloat16)
    ch = c // 2
    flops = 2.0 * b * h * w * (9 * c * 512 + 512 * 512 + 9 * 512 * ch)
    if masks is not None:
        _need(masks, torch.int32, "coupling_bwd_fused masks"); _dense(masks, "coupling_bwd_fused masks")
        if tuple(masks.shape) != (2, bpad * h * w * 32):
            raise ValueError(f"coupling_bwd_fused: masks {tuple(masks.shape)}, expected {(2, bpad * h * w * 32)}")
    _call("nfdp_coupling_bwd_fused_bf16", _ptr(dc3), _ptr(dx), _ptr(packed), _ptr(y1), _ptr(y2), _ptr(dc1), _ptr(dc2), _ptr(dc3col),
          _ptr(masks), b, h, w, c, _stream(), label=f"coupling_bwd_fused<{c},{w}>{' masks' if masks is not None else ''}", flops=flops,
          nbytes=(2048.0 + (256.0 if masks is not None else 2048.0)) * bpad * h * w)
    return dc1, dc2, dc3col


def ldj_add_scaled(ldj: torch.Tensor, v: torch.Tensor, n_sum: int, hw: float):
    _need(ldj, torch.float64, "ldj_add_scaled")
    _call("nfdp_ldj_add_scaled_f32", _ptr(ldj), ldj.numel(), _ptr(v), n_sum, float(hw), _stream())


def ldj_add_levels(ldj: torch.Tensor, terms) -> None:
    """ldj[b] += sum_l hw_l * float(sum(v_l)) for terms = [(v_l fp32 [n_l], hw_l), ...] in ONE launch (nfdp_ldj_add_levels_f32:
    bit-identical to one ldj_add_scaled per level, in order)."""
    _need(ldj, torch.float64, "ldj_add_levels")
    n = len(terms)
    for v, _ in terms:
        _need(v, torch.float32, "ldj_add_levels"); _dense(v, "ldj_add_levels")
    _call("nfdp_ldj_add_levels_f32", _ptr(ldj), ldj.numel(), (ctypes.c_void_p * n)(*[_ptr(v) for v, _ in terms]),
          (ctypes.c_int * n)(*[v.numel() for v, _ in terms]), (ctypes.c_float * n)(*[float(hw) for _, hw in terms]), n, _stream())


def split_prior_squeeze_supported(xh: torch.Tensor) -> bool:
    b, h, w, c = xh.shape
    return bool(xh.is_cuda and xh.dtype == torch.float32 and xh.is_contiguous()
                and lib().nfdp_split_prior_squeeze_supported(c, h, w))


def split_prior_squeeze(xh: torch.Tensor, weight, bias, logs, logp: Optional[torch.Tensor]):
    """Split.transform + the next Squeeze in one launch (nfdp_split_prior_squeeze_f32): xh [B,H,W,C] contiguous ->
    (z [B,H,W,C/2], ynext [B,H/2,W/2,2C]); logp (fp64 [B] or None) += the prior's log-density of z.  weight None: zero mean / log-sd."""
    _need(xh, torch.float32, "split_prior_squeeze"); _dense(xh, "split_prior_squeeze")
    b, h, w, c = xh.shape
    if logp is not None:
        _need(logp, torch.float64, "split_prior_squeeze")
    z = torch.empty((b, h, w, c // 2), device=xh.device, dtype=torch.float32)
    yn = torch.empty((b, h // 2, w // 2, 2 * c), device=xh.device, dtype=torch.float32)
    _call("nfdp_split_prior_squeeze_f32", _ptr(xh), b, h, w, c, _ptr(weight), _ptr(bias), _ptr(logs), _ptr(z), _ptr(yn), _ptr(logp),
          _stream(), label=f"split_prior_squeeze {c} hw{h}", flops=2.0 * b * h * w * 9 * (c // 2) * c, nbytes=float(b * h * w) * c * 8)
    return z, yn


def gauss_logp_prior(zh: torch.Tensor, bias, logs, logp: torch.Tensor) -> None:
    """logp[b] += log N(z_b) under GaussianPrior's spatially constant (mean | log-sd) = bias * exp(3 logs) (nfdp_gauss_logp_prior_f32)."""
    _need(logp, torch.float64, "gauss_logp_prior")
    b, hh, ww, cz = zh.shape
    _call("nfdp_gauss_logp_prior_f32", _ptr(zh), _ld(zh, "gauss_logp_prior"), _ptr(bias), _ptr(logs), b, hh * ww, cz, _ptr(logp), _stream(),
          label="gauss_logp_prior")


def flow_ldj_consts(scale: torch.Tensor, logabsdet: torch.Tensor, hw: float) -> torch.Tensor:
    out = torch.empty(2, device=scale.device, dtype=torch.float32)
    _call("nfdp_flow_ldj_consts_f32", _ptr(scale), scale.numel(), _ptr(logabsdet), float(hw), _ptr(out), _stream())
    return out


def gauss_logp(zh: torch.Tensor, h: torch.Tensor, h_mode: int, logp: torch.Tensor):
    _need(logp, torch.float64, "gauss_logp")
    b, hh, ww, cz = zh.shape
    _call("nfdp_gauss_logp_f32", _ptr(zh), _ld(zh, "gauss_logp"), _ptr(h), h_mode, b, hh * ww, cz, _ptr(logp), _stream())


def gauss_sample(h: torch.Tensor, h_mode: int, eps: torch.Tensor, out: torch.Tensor, temperature: float):
    _dense(eps, "gauss_sample")
    b, hh, ww, cz = out.shape
    _call("nfdp_gauss_sample_f32", _ptr(h), h_mode, _ptr(eps), _ptr(out), _ld(out, "gauss_sample"), b, hh * ww, cz,
                                       float(temperature), _stream())


def zeroconv_bias_only(bias: torch.Tensor, logs: torch.Tensor) -> torch.Tensor:
    h = torch.empty(bias.numel(), device=bias.device, dtype=torch.float32)
    _call("nfdp_zeroconv_bias_only_f32", _ptr(bias), _ptr(logs), _ptr(h), bias.numel(), _stream())
    return h


def bpd_sum(ldj: torch.Tensor, logp: Optional[torch.Tensor], n_bins: float, n_pixel: float) -> torch.Tensor:
    """-> fp64 [2] = (sum_b bpd_b, B)."""
    _need(ldj, torch.float64, "bpd_sum")
    out = torch.empty(2, device=ldj.device, dtype=torch.float64)
    _call("nfdp_bpd_sum_f64", _ptr(ldj), _ptr(logp), ldj.numel(), float(n_bins), float(n_pixel), _ptr(out), _stream())
    return out


def preprocess_u8(u8_nchw: torch.Tensor, noise_nchw: Optional[torch.Tensor], n_bits: int) -> torch.Tensor:
    _need(u8_nchw, torch.uint8, "preprocess_u8"); _dense(u8_nchw, "preprocess_u8")
    b, c, h, w = u8_nchw.shape
    y = torch.empty((b, h, w, c), device=u8_nchw.device, dtype=torch.float32)
    _call("nfdp_preprocess_u8_nchw_to_nhwc_f32", _ptr(u8_nchw), _ptr(noise_nchw), _ptr(y), b, c, h, w, n_bits, _stream())
    return y


def preprocess_f32(x: torch.Tensor, noise: Optional[torch.Tensor], n_bits: int) -> torch.Tensor:
    _need(x, torch.float32, "preprocess_f32")
    x = x.contiguous()
    y = torch.empty_like(x)
    _call("nfdp_preprocess_f32", _ptr(x), _ptr(noise), _ptr(y), x.numel(), n_bits, _stream())
    return y


def postprocess_u8(xh: torch.Tensor, n_bins: float) -> torch.Tensor:
    _need(xh, torch.float32, "postprocess_u8"); _dense(xh, "postprocess_u8")
    b, h, w, c = xh.shape
    u8 = torch.empty((b, c, h, w), device=xh.device, dtype=torch.uint8)
    _call("nfdp_postprocess_nhwc_to_u8_nchw", _ptr(xh), _ptr(u8), b, c, h, w, float(n_bins), _stream())
    return u8


# ----------------------------------------------------------------------------- diffusion prior: U-Net blocks
def ws_standardize(w: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    _need(w, torch.float32, "ws_standardize"); _dense(w, "ws_standardize")
    out = torch.empty_like(w)
    _call("nfdp_ws_standardize_f32", _ptr(w), _ptr(out), w.shape[0], w.numel() // w.shape[0], float(eps), _stream())
    return out


def groupnorm_silu(xh: torch.Tensor, groups: int, gamma, beta, scale_shift: Optional[torch.Tensor], eps: float = 1e-5,
                   residual: Optional[torch.Tensor] = None):
    _need(xh, torch.float32, "groupnorm_silu"); _dense(xh, "groupnorm_silu")
    if residual is not None:
        _need(residual, torch.float32, "groupnorm_silu residual"); _dense(residual, "groupnorm_silu residual")
        assert residual.shape == xh.shape
    b, h, w, c = xh.shape
    y = torch.empty_like(xh)
    ss_ld = 0
    if scale_shift is not None:                       # [B, 2C] rows, possibly a column slice of a wider [B, sum] buffer
        assert scale_shift.shape == (b, 2 * c) and scale_shift.stride(1) == 1
        ss_ld = scale_shift.stride(0) if b > 1 else 2 * c
    _call("nfdp_groupnorm_silu_res_f32", _ptr(xh), _ptr(y), b, h * w, c, groups, _ptr(gamma), _ptr(beta), _ptr(scale_shift),
          ss_ld, float(eps), _ptr(residual), _stream(), label="groupnorm_silu_kernel", nbytes=8.0 * xh.numel())
    return y


_CONV_GN_FUSED = True


def set_conv_gn_fused(on: bool) -> None:
    """U-Net Block (WS-conv -> GroupNorm -> scale/shift -> SiLU [-> + residual]) as one launch where the geometry allows (default) or as
    conv + GroupNorm launches."""
    global _CONV_GN_FUSED
    _CONV_GN_FUSED = bool(on)


def conv2d_gn_silu_supported(xh: torch.Tensor, cout: int, k: int, groups: int, *vectors, extra_cin: int = 0) -> bool:
    """The one-launch Block is available for this input (fp32 NHWC, no grad) and these parameter vectors (16-byte aligned)?
    extra_cin: channels of a second source (the input is then the concatenation, see conv2d_gn_silu's x2)."""
    if not (_CONV_GN_FUSED and xh.is_cuda and xh.dtype == torch.float32 and xh.dim() == 4 and xh.stride(3) == 1):
        return False
    b, h, w, cin = xh.shape
    cin += extra_cin
    if any(v is not None and v.data_ptr() % 16 != 0 for v in vectors) or xh.data_ptr() % 16 != 0 or xh.stride(2) % 4 != 0:
        return False
    return bool(lib().nfdp_conv2d_gn_silu_supported(b, h, w, cin, cout, k, groups))


def conv2d_gn_silu_shape_supported(b: int, h: int, w: int, cin: int, cout: int, k: int, groups: int) -> bool:
    """conv2d_gn_silu_supported for a fresh (aligned, dense) input of this shape."""
    return bool(_CONV_GN_FUSED and lib().nfdp_conv2d_gn_silu_supported(b, h, w, cin, cout, k, groups))


def cat_sources_ok(x1: torch.Tensor, x2: torch.Tensor) -> bool:
    """Can a conv read torch.cat((x1, x2), dim=-1) from the two tensors (nfdp_conv2d_cat_nhwc_f32 / _gn_silu_cat_)?"""
    if not (_CONV_CAT and x1.is_cuda and x1.dtype == torch.float32 and x2.dtype == torch.float32 and x1.dim() == 4 and x2.dim() == 4):
        return False
    if x1.shape[:3] != x2.shape[:3] or x1.shape[3] % 8 != 0 or x2.shape[3] % 8 != 0 or x1.stride(3) != 1 or x2.stride(3) != 1:
        return False
    return all(t.data_ptr() % 16 == 0 and t.stride(2) % 4 == 0 for t in (x1, x2))


_CONV_CAT = True


def set_conv_cat(on: bool) -> None:
    """Two-source convs for the U-Net's skip connections (default on); off: concat_channels + the one-source conv (A/B runs, tests)."""
    global _CONV_CAT
    _CONV_CAT = bool(on)


def conv2d_cat(x1: torch.Tensor, x2: torch.Tensor, packed: torch.Tensor, cout: int, k: int, add=None) -> torch.Tensor:
    """conv2d(torch.cat((x1, x2), dim=-1)) in exact fp32 without the concatenated tensor; check cat_sources_ok first."""
    b, h, w, c1 = x1.shape
    c2 = x2.shape[3]
    out = torch.empty((b, h, w, cout), device=x1.device, dtype=torch.float32)
    _call("nfdp_conv2d_cat_nhwc_f32", _ptr(x1), _ld(x1, "conv2d_cat"), c1, _ptr(x2), _ld(x2, "conv2d_cat"), c2, _ptr(packed), _ptr(out), cout,
          b, h, w, cout, k, _ptr(add), _f32_compute(F32), _stream(), label=f"conv2d_cat k{k} {c1}+{c2}->{cout} hw{h}", flops=2.0 * b * h * w * (c1 + c2) * k * k * cout)
    return out


def conv2d_gn_silu(xh: torch.Tensor, packed: torch.Tensor, cout: int, k: int, bias, groups: int, gamma, beta,
                   scale_shift: Optional[torch.Tensor], eps: float = 1e-5, residual: Optional[torch.Tensor] = None,
                   x2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Block of the U-Net in one launch (nfdp_conv2d_gn_silu_nhwc_f32); check conv2d_gn_silu_supported first.  x2: the input is
    torch.cat((xh, x2), dim=-1), read from the two tensors (cat_sources_ok)."""
    _need(xh, torch.float32, "conv2d_gn_silu")
    b, h, w, cin = xh.shape
    y = torch.empty((b, h, w, cout), device=xh.device, dtype=torch.float32)
    ss_ld = 0
    if scale_shift is not None:
        assert scale_shift.shape == (b, 2 * cout) and scale_shift.stride(1) == 1
        ss_ld = scale_shift.stride(0) if b > 1 else 2 * cout
    res_ld = 0
    if residual is not None:
        _need(residual, torch.float32, "conv2d_gn_silu residual"); _dense(residual, "conv2d_gn_silu residual")
        assert residual.shape == y.shape
        res_ld = cout
    if x2 is not None:
        c2 = x2.shape[3]
        _call("nfdp_conv2d_gn_silu_cat_nhwc_f32", _ptr(xh), _ld(xh, "conv2d_gn_silu"), cin, _ptr(x2), _ld(x2, "conv2d_gn_silu"), c2,
              _ptr(packed), _ptr(y), cout, b, h, w, cout, k, _ptr(bias), groups, _ptr(gamma), _ptr(beta), _ptr(scale_shift), ss_ld,
              float(eps), _ptr(residual), res_ld, _f32_compute(F32), _stream(),
              label=f"conv_splitk_f32+gn cat k{k} {cin}+{c2}->{cout} hw{h}", flops=2.0 * b * h * w * (cin + c2) * k * k * cout)
        return y
    _call("nfdp_conv2d_gn_silu_nhwc_f32", _ptr(xh), _ld(xh, "conv2d_gn_silu"), _ptr(packed), _ptr(y), cout, b, h, w, cin, cout, k,
          _ptr(bias), groups, _ptr(gamma), _ptr(beta), _ptr(scale_shift), ss_ld, float(eps), _ptr(residual), res_ld, _f32_compute(F32),
          _stream(), label=f"conv_splitk_f32+gn k{k} {cin}->{cout} hw{h}", flops=2.0 * b * h * w * cin * k * k * cout)
    return y


_CONV_GN_CARRIED = True


def set_conv_gn_carried(on: bool) -> None:
    """ResnetBlock: block1's GroupNorm inside block2's conv where a conv tile is part of an image (default) or as a launch of its own."""
    global _CONV_GN_CARRIED
    _CONV_GN_CARRIED = bool(on)


def conv2d_gn_carried_supported(xh: torch.Tensor, cmid: int, k: int, groups1: int, groups2: int, fused2: bool, *vectors,
                                extra_cin: int = 0) -> bool:
    """Both halves of the carried GroupNorm are available: conv (xh [+ extra_cin channels of a second source]) -> cmid with group sums
    (nfdp_conv2d_stats_*), then the 3x3 conv cmid -> cmid that normalises on load (nfdp_conv2d_gnin_*; fused2: with its own Block
    epilogue).  fp32 NHWC, no grad, split-product fp32 mode, 16-byte aligned vectors."""
    if not (_CONV_GN_CARRIED and _F32_PRODUCTS == "bf16x6" and xh.is_cuda and xh.dtype == torch.float32 and xh.dim() == 4 and
            xh.stride(3) == 1 and not torch.is_grad_enabled()):
        return False
    if xh.data_ptr() % 16 or xh.stride(2) % 4 or any(v is not None and v.data_ptr() % 16 for v in vectors):
        return False
    b, h, w, cin = xh.shape
    if not lib().nfdp_conv2d_stats_tile(b, h, w, cin + extra_cin, cmid, k, groups1):
        return False
    return bool(lib().nfdp_conv2d_gnin_supported(b, h, w, cmid, cmid, groups1, groups2 if fused2 else 0))


_CONV_RES_FUSED = True


def set_conv_res_fused(on: bool) -> None:
    """ResnetBlock: the 1x1 res_conv in the launch of block1's 3x3 conv (default, where conv2d_res_supported) or as a launch of its own."""
    global _CONV_RES_FUSED
    _CONV_RES_FUSED = bool(on)


def conv2d_res_supported(xh: torch.Tensor, cout: int, *vectors, extra_cin: int = 0) -> bool:
    """conv2d_stats(..., res=...) is available: the 3x3 conv of this input takes the LDS-staged kernel and the 1x1 weight has a split image."""
    if not (_CONV_RES_FUSED and _F32_PRODUCTS == "bf16x6" and xh.is_cuda and xh.dtype == torch.float32 and xh.dim() == 4 and
            xh.stride(3) == 1 and not torch.is_grad_enabled()):
        return False
    if xh.data_ptr() % 16 or xh.stride(2) % 4 or any(v is not None and v.data_ptr() % 16 for v in vectors):
        return False
    b, h, w, cin = xh.shape
    return bool(lib().nfdp_conv2d_res_supported(b, h, w, cin + extra_cin, cout))


def conv2d_stats(xh: torch.Tensor, packed: torch.Tensor, cout: int, k: int, bias, groups: int, x2: Optional[torch.Tensor] = None, res=None):
    """conv2d (of torch.cat((xh, x2), -1) when x2 is given) that also leaves the fp64 (sum, sum of squares) of every (pixel tile, group) of
    its output (nfdp_conv2d_stats_nhwc_f32) -> (y, sums [tiles, groups, 2], pixels per tile).  res = (packed 1x1 weight, bias): the 1x1
    conv of the same input in the same launch (nfdp_conv2d_res_nhwc_f32; k = 3, check conv2d_res_supported) -> (y, sums, tile, y_res)."""
    _need(xh, torch.float32, "conv2d_stats")
    b, h, w, c1 = xh.shape
    c2 = 0 if x2 is None else x2.shape[3]
    tile = int(lib().nfdp_conv2d_stats_tile(b, h, w, c1 + c2, cout, k, groups))
    if tile <= 0:
        raise ValueError(f"conv2d_stats: unsupported geometry {tuple(xh.shape)} (+{c2}) -> {cout}, k={k}, groups={groups}")
    y = torch.empty((b, h, w, cout), device=xh.device, dtype=torch.float32)
    sums = torch.empty((b * h * w // tile, groups, 2), device=xh.device, dtype=torch.float64)
    if res is not None:
        assert k == 3
        yr = torch.empty((b, h, w, cout), device=xh.device, dtype=torch.float32)
        _call("nfdp_conv2d_res_nhwc_f32", _ptr(xh), _ld(xh, "conv2d_stats"), c1, _ptr(x2), 0 if x2 is None else _ld(x2, "conv2d_stats"), c2,
              _ptr(packed), _ptr(y), cout, b, h, w, cout, _ptr(bias), groups, _ptr(sums), _ptr(res[0]), _ptr(res[1]), _ptr(yr), cout, _stream(),
              label=f"conv3x3_ldsx+sums+res{' cat' if c2 else ''} {c1}{'+' + str(c2) if c2 else ''}->{cout} hw{h}",
              flops=2.0 * b * h * w * (c1 + c2) * 10 * cout)
        return y, sums, tile, yr
    _call("nfdp_conv2d_stats_nhwc_f32", _ptr(xh), _ld(xh, "conv2d_stats"), c1, _ptr(x2), 0 if x2 is None else _ld(x2, "conv2d_stats"), c2,
          _ptr(packed), _ptr(y), cout, b, h, w, cout, k, _ptr(bias), groups, _ptr(sums), _f32_compute(F32), _stream(),
          label=f"conv_splitk_f32+sums{' cat' if c2 else ''} k{k} {c1}{'+' + str(c2) if c2 else ''}->{cout} hw{h}", flops=2.0 * b * h * w * (c1 + c2) * k * k * cout)
    return y, sums, tile


def conv2d_gnin(xh: torch.Tensor, sums: torch.Tensor, tile: int, in_groups: int, in_gamma, in_beta, in_scale_shift, packed: torch.Tensor,
                cout: int, bias, in_eps: float = 1e-5, out_groups: int = 0, out_gamma=None, out_beta=None, out_scale_shift=None,
                out_eps: float = 1e-5, residual: Optional[torch.Tensor] = None, sums_groups: int = 0):
    """3x3 conv over SiLU(GroupNorm(xh) * (scale + 1) + shift), xh the raw output and `sums` the group sums of conv2d_stats
    (nfdp_conv2d_gnin_nhwc_f32); out_groups > 0: followed by the conv's own Block epilogue (+ residual).  sums_groups > 0 (with
    out_groups = 0): also the group sums of ITS output -> (y, sums, tile) for groupnorm_silu_sums."""
    _need(xh, torch.float32, "conv2d_gnin"); _need(sums, torch.float64, "conv2d_gnin")
    b, h, w, cin = xh.shape
    y = torch.empty((b, h, w, cout), device=xh.device, dtype=torch.float32)

    def _ss_ld(ss, c):
        if ss is None:
            return 0
        assert ss.shape == (b, 2 * c) and ss.stride(1) == 1
        return ss.stride(0) if b > 1 else 2 * c
    res_ld = 0
    if residual is not None:
        _need(residual, torch.float32, "conv2d_gnin residual"); _dense(residual, "conv2d_gnin residual")
        assert residual.shape == y.shape
        res_ld = cout
    osums, otile = None, 0
    if sums_groups:
        otile = int(lib().nfdp_conv2d_stats_tile(b, h, w, cin, cout, 3, sums_groups))
        assert otile > 0 and out_groups == 0
        osums = torch.empty((b * h * w // otile, sums_groups, 2), device=xh.device, dtype=torch.float64)
    _call("nfdp_conv2d_gnin_nhwc_f32", _ptr(xh), _ld(xh, "conv2d_gnin"), _ptr(sums), int(tile), in_groups, _ptr(in_gamma), _ptr(in_beta),
          _ptr(in_scale_shift), _ss_ld(in_scale_shift, cin), float(in_eps), _ptr(packed), _ptr(y), cout, b, h, w, cin, cout, _ptr(bias),
          out_groups, _ptr(out_gamma), _ptr(out_beta), _ptr(out_scale_shift), _ss_ld(out_scale_shift, cout), float(out_eps), _ptr(residual),
          res_ld, sums_groups, _ptr(osums), _stream(),
          label=f"conv3x3_ldsx gn-in{'+gn' if out_groups else ''}{'+sums' if sums_groups else ''} {cin}->{cout} hw{h}", flops=2.0 * b * h * w * cin * 9 * cout)
    return (y, osums, otile) if sums_groups else y


_GN_FROM_SUMS = True
GN_ON_LOAD_MAX_TILES = 256         # ResnetBlock: block1's norm rides in block2's conv up to this many 64 x 64 output tiles (one wave of the chip)


def set_groupnorm_from_sums(on: bool) -> None:
    """Block 2's GroupNorm from the group sums its conv left (streaming kernel, default) or with its own reductions."""
    global _GN_FROM_SUMS
    _GN_FROM_SUMS = bool(on)


def groupnorm_silu_sums_supported(b: int, h: int, w: int, cin: int, c: int, groups: int, *vectors) -> bool:
    """Block 2 (3x3 conv cin -> c on the normalise-on-load kernel, then GroupNorm) can take its norm's statistics from the conv's sums?"""
    if not _GN_FROM_SUMS or any(v is not None and v.data_ptr() % 16 for v in vectors):
        return False
    tile = int(lib().nfdp_conv2d_stats_tile(b, h, w, cin, c, 3, groups))
    return tile > 0 and bool(lib().nfdp_groupnorm_silu_sums_supported(b, h * w, c, groups, tile))


def groupnorm_silu_sums(xh: torch.Tensor, sums: torch.Tensor, tile: int, groups: int, gamma, beta, scale_shift: Optional[torch.Tensor],
                        eps: float = 1e-5, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """groupnorm_silu with the statistics from the producing conv's per-tile group sums (nfdp_groupnorm_silu_sums_f32)."""
    _need(xh, torch.float32, "groupnorm_silu_sums"); _dense(xh, "groupnorm_silu_sums"); _need(sums, torch.float64, "groupnorm_silu_sums")
    b, h, w, c = xh.shape
    if residual is not None:
        _need(residual, torch.float32, "groupnorm_silu_sums residual"); _dense(residual, "groupnorm_silu_sums residual")
        assert residual.shape == xh.shape
    y = torch.empty_like(xh)
    ss_ld = 0
    if scale_shift is not None:
        assert scale_shift.shape == (b, 2 * c) and scale_shift.stride(1) == 1
        ss_ld = scale_shift.stride(0) if b > 1 else 2 * c
    _call("nfdp_groupnorm_silu_sums_f32", _ptr(xh), _ptr(y), _ptr(sums), int(tile), b, h * w, c, groups, _ptr(gamma), _ptr(beta), _ptr(scale_shift),
          ss_ld, float(eps), _ptr(residual), _stream(), label="groupnorm_silu_sums_kernel", nbytes=8.0 * xh.numel())
    return y


_CONV_K7_NARROW = True


def set_conv_k7_narrow(on: bool) -> None:
    """Unet.init_conv over 1-4 input channels on the direct fp32 kernel (default) or on the implicit-GEMM conv."""
    global _CONV_K7_NARROW
    _CONV_K7_NARROW = bool(on)


def conv_k7_narrow_supported(xh: torch.Tensor, cout: int, bias=None) -> bool:
    if not (_CONV_K7_NARROW and xh.is_cuda and xh.dtype == torch.float32 and xh.dim() == 4 and (xh.stride(3) == 1 or xh.shape[3] == 1) and
            not torch.is_grad_enabled() and _is_dense_pixels(xh)):
        return False
    b, h, w, cin = xh.shape
    return (bias is None or bias.data_ptr() % 16 == 0) and bool(lib().nfdp_conv_k7_narrow_supported(b, h, w, cin, cout))


def conv_k7_pack(w_oihw: torch.Tensor) -> torch.Tensor:
    """[Cout][Cin][7][7] -> the [49][Cin][Cout] copy conv_k7_narrow reads (nfdp_conv_k7_pack_f32)."""
    _need(w_oihw, torch.float32, "conv_k7_pack"); _dense(w_oihw, "conv_k7_pack")
    cout, cin = w_oihw.shape[:2]
    assert tuple(w_oihw.shape[2:]) == (7, 7)
    out = torch.empty(cout * cin * 49, device=w_oihw.device, dtype=torch.float32)
    _call("nfdp_conv_k7_pack_f32", _ptr(w_oihw), _ptr(out), cout, cin, _stream())
    return out


def conv_k7_narrow(xh: torch.Tensor, w_k7: torch.Tensor, cout: int, bias=None) -> torch.Tensor:
    """7x7 conv (padding 3) over 1-4 input channels as exact fp32 FMA chains (nfdp_conv_k7_narrow_f32); w_k7 = conv_k7_pack(weight)."""
    b, h, w, cin = xh.shape
    assert w_k7.numel() == cout * cin * 49
    y = torch.empty((b, h, w, cout), device=xh.device, dtype=torch.float32)
    _call("nfdp_conv_k7_narrow_f32", _ptr(xh), _ld(xh, "conv_k7_narrow"), _ptr(w_k7), _ptr(bias), _ptr(y), cout, b, h, w, cin, cout, _stream(),
          label=f"conv_k7_narrow {cin}->{cout} hw{h}", flops=2.0 * b * h * w * cin * 49 * cout)
    return y


_CONV_UP_FUSED = True


def set_conv_up_fused(on: bool) -> None:
    """Upsample (nearest x2 -> 3x3 conv) as one launch where conv2d_up2_supported (default) or as two."""
    global _CONV_UP_FUSED
    _CONV_UP_FUSED = bool(on)


def conv2d_up2_supported(xh: torch.Tensor, cout: int, bias=None) -> bool:
    if not (_CONV_UP_FUSED and _F32_PRODUCTS == "bf16x6" and xh.is_cuda and xh.dtype == torch.float32 and xh.dim() == 4 and
            xh.stride(3) == 1 and not torch.is_grad_enabled()):
        return False
    if xh.data_ptr() % 16 or xh.stride(2) % 4 or (bias is not None and bias.data_ptr() % 16):
        return False
    b, h, w, cin = xh.shape
    return _is_dense_pixels(xh) and bool(lib().nfdp_conv2d_up2_supported(b, 2 * h, 2 * w, cin, cout))


def _is_dense_pixels(xh: torch.Tensor) -> bool:
    """Pixels at one stride (row = pixel index * stride(2)): what a flat pixel index into [B, H, W] needs."""
    b, h, w, _ = xh.shape
    return xh.stride(1) == w * xh.stride(2) and xh.stride(0) == h * xh.stride(1)


def conv2d_up2(xh: torch.Tensor, packed: torch.Tensor, cout: int, bias=None) -> torch.Tensor:
    """conv3x3(nearest-x2 upsampling of xh) without the upsampled tensor (nfdp_conv2d_up2_nhwc_f32; check conv2d_up2_supported)."""
    b, h, w, cin = xh.shape
    y = torch.empty((b, 2 * h, 2 * w, cout), device=xh.device, dtype=torch.float32)
    _call("nfdp_conv2d_up2_nhwc_f32", _ptr(xh), _ld(xh, "conv2d_up2"), _ptr(packed), _ptr(y), cout, b, 2 * h, 2 * w, cin, cout, _ptr(bias),
          _stream(), label=f"conv3x3_ldsx up2 {cin}->{cout} hw{2 * h}", flops=2.0 * b * 4 * h * w * cin * 9 * cout)
    return y


_CONV_SQ_FUSED = True


def set_conv_squeeze_fused(on: bool) -> None:
    """Downsample (2x2 squeeze -> 1x1 conv) as one launch where conv2d_squeeze_supported (default) or as two."""
    global _CONV_SQ_FUSED
    _CONV_SQ_FUSED = bool(on)


def conv2d_squeeze_supported(xh: torch.Tensor, cout: int, bias=None) -> bool:
    if not (_CONV_SQ_FUSED and xh.is_cuda and xh.dtype == torch.float32 and xh.dim() == 4 and xh.stride(3) == 1 and
            not torch.is_grad_enabled()):
        return False
    b, h, w, c = xh.shape
    if xh.data_ptr() % 16 or xh.stride(2) % 4 or (bias is not None and bias.data_ptr() % 16) or h % 2 or w % 2 or not _is_dense_pixels(xh):
        return False
    return bool(lib().nfdp_conv2d_squeeze_supported(b, h // 2, w // 2, c, cout))


def conv2d_squeeze(xh: torch.Tensor, packed2x2: torch.Tensor, cout: int, bias=None) -> torch.Tensor:
    """conv1x1(squeeze(xh)) as the 2x2 / stride-2 conv it is (nfdp_conv2d_squeeze_nhwc_f32); packed2x2 = conv_pack(w.view(cout, C, 2, 2), F32)."""
    b, h, w, c = xh.shape
    y = torch.empty((b, h // 2, w // 2, cout), device=xh.device, dtype=torch.float32)
    _call("nfdp_conv2d_squeeze_nhwc_f32", _ptr(xh), _ld(xh, "conv2d_squeeze"), _ptr(packed2x2), _ptr(y), cout, b, h // 2, w // 2, c, cout,
          _ptr(bias), _f32_compute(F32), _stream(), label=f"conv_splitk_f32 squeeze {c}x4->{cout} hw{h // 2}", flops=2.0 * b * h * w * c * cout)
    return y


def conv2d_gn_silu_res_supported(xh: torch.Tensor, cout: int, groups: int, *vectors, extra_cin: int = 0) -> bool:
    """conv2d_gn_silu_res is available (the one-launch Block on the LDS-staged kernel + a split image for the 1x1 weight)?"""
    if not (_CONV_RES_FUSED and _F32_PRODUCTS == "bf16x6" and conv2d_gn_silu_supported(xh, cout, 3, groups, *vectors, extra_cin=extra_cin)):
        return False
    b, h, w, cin = xh.shape
    return bool(lib().nfdp_conv2d_gn_silu_res_supported(b, h, w, cin + extra_cin, cout, groups))


def conv2d_gn_silu_res(xh: torch.Tensor, packed: torch.Tensor, cout: int, bias, groups: int, gamma, beta, scale_shift, packed_res: torch.Tensor,
                       bias_res, eps: float = 1e-5, x2: Optional[torch.Tensor] = None):
    """Block (3x3 WS-conv -> GroupNorm -> scale/shift -> SiLU) and the 1x1 res_conv of the same input in one launch
    (nfdp_conv2d_gn_silu_res_nhwc_f32) -> (h, res)."""
    b, h, w, c1 = xh.shape
    c2 = 0 if x2 is None else x2.shape[3]
    y = torch.empty((b, h, w, cout), device=xh.device, dtype=torch.float32)
    yr = torch.empty((b, h, w, cout), device=xh.device, dtype=torch.float32)
    ss_ld = 0
    if scale_shift is not None:
        assert scale_shift.shape == (b, 2 * cout) and scale_shift.stride(1) == 1
        ss_ld = scale_shift.stride(0) if b > 1 else 2 * cout
    _call("nfdp_conv2d_gn_silu_res_nhwc_f32", _ptr(xh), _ld(xh, "conv2d_gn_silu_res"), c1, _ptr(x2), 0 if x2 is None else _ld(x2, "conv2d_gn_silu_res"),
          c2, _ptr(packed), _ptr(y), cout, b, h, w, cout, _ptr(bias), groups, _ptr(gamma), _ptr(beta), _ptr(scale_shift), ss_ld, float(eps),
          _ptr(packed_res), _ptr(bias_res), _ptr(yr), cout, _stream(),
          label=f"conv_splitk_f32+gn+res{' cat' if c2 else ''} {c1}{'+' + str(c2) if c2 else ''}->{cout} hw{h}", flops=2.0 * b * h * w * (c1 + c2) * 10 * cout)
    return y, yr


def chan_layernorm(xh: torch.Tensor, gain: torch.Tensor, residual: Optional[torch.Tensor] = None, eps: float = 1e-5):
    _need(xh, torch.float32, "chan_layernorm"); _dense(xh, "chan_layernorm")
    c = xh.shape[-1]
    y = torch.empty_like(xh)
    _call("nfdp_chan_layernorm_f32", _ptr(xh), _ptr(y), xh.numel() // c, c, _ptr(gain), float(eps), _ptr(residual), _stream(),
          label="chan_layernorm_kernel", nbytes=8.0 * xh.numel())
    return y


_CONV_LN_FUSED = True


def set_conv_ln_fused(on: bool) -> None:
    """The attention blocks' 1x1 convs with their channel LayerNorms in the same launch (default) or as separate launches."""
    global _CONV_LN_FUSED
    _CONV_LN_FUSED = bool(on)


def conv1x1_ln_supported(xh: torch.Tensor, cout: int, pre: bool, post: bool, *vectors) -> bool:
    """nfdp_conv1x1_ln_f32 takes this input (fp32 NHWC, no grad, split-product fp32 mode) and these vectors (16-byte aligned)?"""
    if not (_CONV_LN_FUSED and _F32_PRODUCTS == "bf16x6" and xh.is_cuda and xh.dtype == torch.float32 and xh.dim() == 4 and
            xh.stride(3) == 1 and not torch.is_grad_enabled()):
        return False
    if xh.data_ptr() % 16 or xh.stride(2) % 4 or any(v is not None and v.data_ptr() % 16 for v in vectors):
        return False
    b, h, w, cin = xh.shape
    return bool(lib().nfdp_conv1x1_ln_supported(b * h * w, cin, cout, int(pre), int(post)))


def conv1x1_ln(xh: torch.Tensor, packed: torch.Tensor, cout: int, bias=None, pre_gain=None, post_gain=None, residual=None,
               eps: float = 1e-5) -> torch.Tensor:
    """1x1 conv of LayerNorm(xh) (pre_gain) and / or LayerNorm of the conv's output (post_gain) + residual, in one launch
    (nfdp_conv1x1_ln_f32; check conv1x1_ln_supported first).  packed: conv_pack(w, F32)."""
    b, h, w, cin = xh.shape
    y = torch.empty((b, h, w, cout), device=xh.device, dtype=torch.float32)
    res_ld = 0
    if residual is not None:
        _need(residual, torch.float32, "conv1x1_ln residual"); _dense(residual, "conv1x1_ln residual")
        assert residual.shape == y.shape
        res_ld = cout
    _call("nfdp_conv1x1_ln_f32", _ptr(xh), _ld(xh, "conv1x1_ln"), _ptr(packed), _ptr(y), cout, b * h * w, cin, cout, _ptr(bias),
          _ptr(pre_gain), _ptr(post_gain), _ptr(residual), res_ld, float(eps), _stream(),
          label=f"conv1x1_ln{'+pre' if pre_gain is not None else ''}{'+post' if post_gain is not None else ''}"
                f"{'+res' if residual is not None and post_gain is None else ''} {cin}->{cout}",
          flops=2.0 * b * h * w * cin * cout)
    return y


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], pre_act: int = 0, post_act: int = 0):
    """x [B,I] -> [B,O]; pre_act 1 = SiLU; post_act 1 = GELU(erf), 2 = SiLU."""
    _need(x, torch.float32, "linear"); _dense(x, "linear"); _dense(w, "linear")
    b, i = x.shape
    o = w.shape[0]
    out = torch.empty((b, o), device=x.device, dtype=torch.float32)
    _call("nfdp_linear_f32", _ptr(x), _ptr(w), _ptr(bias), _ptr(out), b, i, o, pre_act, post_act, _stream())
    return out


def sinusoidal_emb(t: torch.Tensor, dim: int) -> torch.Tensor:
    _need(t, torch.int64, "sinusoidal_emb")
    out = torch.empty((t.numel(), dim), device=t.device, dtype=torch.float32)
    _call("nfdp_sinusoidal_emb_f32", _ptr(t.contiguous()), _ptr(out), t.numel(), dim, _stream())
    return out


def learned_sinusoidal_emb(t: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """RandomOrLearnedSinusoidalPosEmb.forward: t int64 [B], weights fp32 [half] -> [B, 2*half + 1]."""
    _need(t, torch.int64, "learned_sinusoidal_emb"); _need(weights, torch.float32, "learned_sinusoidal_emb"); _dense(weights, "learned_sinusoidal_emb")
    half = weights.numel()
    out = torch.empty((t.numel(), 2 * half + 1), device=t.device, dtype=torch.float32)
    _call("nfdp_learned_sinusoidal_emb_f32", _ptr(t.contiguous()), _ptr(weights), _ptr(out), t.numel(), half, _stream())
    return out


def learned_sinusoidal_bwd(t: torch.Tensor, weights: torch.Tensor, dout: torch.Tensor, dw: Optional[torch.Tensor] = None,
                           accumulate: bool = False) -> torch.Tensor:
    """Gradient of learned_sinusoidal_emb with respect to `weights` (dout [B, 2*half + 1])."""
    _need(t, torch.int64, "learned_sinusoidal_bwd"); _need(dout, torch.float32, "learned_sinusoidal_bwd"); _dense(dout, "learned_sinusoidal_bwd")
    half = weights.numel()
    if dout.shape[-1] != 2 * half + 1 or dout.numel() != t.numel() * (2 * half + 1):
        raise ValueError("learned_sinusoidal_bwd: dout must be [B, 2*half + 1]")
    if dw is None:
        dw, accumulate = torch.empty(half, device=t.device, dtype=torch.float32), False
    _call("nfdp_learned_sinusoidal_bwd_f32", _ptr(t.contiguous()), _ptr(weights), _ptr(dout), _ptr(dw), t.numel(), half, int(accumulate),
          _stream())
    return dw


def linear_attention(qkv: torch.Tensor, heads: int, dim_head: int, scale: float) -> torch.Tensor:
    _need(qkv, torch.float32, "linear_attention"); _dense(qkv, "linear_attention")
    b, h, w, _ = qkv.shape
    out = torch.empty((b, h, w, heads * dim_head), device=qkv.device, dtype=torch.float32)
    nbytes = lib().nfdp_linear_attention_ws_bytes(b, h * w, heads)
    if nbytes:                                                       # n > 256: chunked over the grid, two launches
        ws = torch.empty(nbytes, device=qkv.device, dtype=torch.uint8)
        _call("nfdp_linear_attention_chunked_f32", _ptr(qkv), _ptr(out), b, h * w, heads, dim_head, float(scale), _ptr(ws), _stream(),
              label="linear_attention_chunked")
        return out
    _call("nfdp_linear_attention_f32", _ptr(qkv), _ptr(out), b, h * w, heads, dim_head, float(scale), _stream(),
          label="linear_attention_kernel")
    return out


_ATTN_QKV_FUSED = True


def set_attention_qkv_fused(on: bool) -> None:
    """PreNorm -> LinearAttention with the to_qkv projection inside the attention launch (default, where supported) or as two launches."""
    global _ATTN_QKV_FUSED
    _ATTN_QKV_FUSED = bool(on)


def linear_attention_qkv_supported(xh: torch.Tensor, heads: int, dim_head: int, gain, linear: bool = True) -> bool:
    """to_qkv inside the attention core's launch is available for this input (linear: LinearAttention, else Attention)?"""
    if not (_ATTN_QKV_FUSED and _CONV_LN_FUSED and _F32_PRODUCTS == "bf16x6" and xh.is_cuda and xh.dtype == torch.float32 and xh.dim() == 4 and
            xh.stride(3) == 1 and not torch.is_grad_enabled() and _is_dense_pixels(xh)):
        return False
    if xh.data_ptr() % 16 or xh.stride(2) % 4 or gain.data_ptr() % 16:
        return False
    b, h, w, c = xh.shape
    fn = lib().nfdp_linear_attention_qkv_supported if linear else lib().nfdp_attention_qkv_supported
    return bool(fn(h * w, c, heads, dim_head))


def linear_attention_qkv(xh: torch.Tensor, gain: torch.Tensor, packed: torch.Tensor, heads: int, dim_head: int, scale: float,
                         eps: float = 1e-5, linear: bool = True) -> torch.Tensor:
    """LinearAttention(LayerNorm(xh)) core (linear=False: Attention's) with its to_qkv projection in the same launch
    (nfdp_linear_attention_qkv_f32 / nfdp_attention_qkv_f32; check linear_attention_qkv_supported first).  packed: conv_pack(to_qkv.weight, F32)."""
    b, h, w, c = xh.shape
    out = torch.empty((b, h, w, heads * dim_head), device=xh.device, dtype=torch.float32)
    if not linear:
        _call("nfdp_attention_qkv_f32", _ptr(xh), _ld(xh, "attention_qkv"), _ptr(gain), _ptr(packed), _ptr(out), b, h * w, c, heads,
              dim_head, float(scale), float(eps), _stream(), label=f"attention+qkv {c} hw{h}", flops=2.0 * b * h * w * c * 3 * heads * dim_head)
        return out
    nbytes = lib().nfdp_linear_attention_qkv_ws_bytes(b, h * w, heads)
    ws = torch.empty(nbytes, device=xh.device, dtype=torch.uint8) if nbytes else None   # n > 256: chunked over the grid, two launches
    _call("nfdp_linear_attention_qkv_f32", _ptr(xh), _ld(xh, "linear_attention_qkv"), _ptr(gain), _ptr(packed), _ptr(out), b, h * w, c, heads,
          dim_head, float(scale), float(eps), _ptr(ws), _stream(), label=f"linear_attention+qkv {c} hw{h}", flops=2.0 * b * h * w * c * 3 * heads * dim_head)
    return out


def attention(qkv: torch.Tensor, heads: int, dim_head: int, scale: float) -> torch.Tensor:
    _need(qkv, torch.float32, "attention"); _dense(qkv, "attention")
    b, h, w, _ = qkv.shape
    out = torch.empty((b, h, w, heads * dim_head), device=qkv.device, dtype=torch.float32)
    _call("nfdp_attention_f32", _ptr(qkv), _ptr(out), b, h * w, heads, dim_head, float(scale), _stream(), label="attention_kernel")
    return out


def upsample2x(xh: torch.Tensor) -> torch.Tensor:
    _need(xh, torch.float32, "upsample2x"); _dense(xh, "upsample2x")
    b, h, w, c = xh.shape
    y = torch.empty((b, 2 * h, 2 * w, c), device=xh.device, dtype=torch.float32)
    _call("nfdp_upsample_nearest2x_f32", _ptr(xh), _ptr(y), b, c, h, w, _stream())
    return y


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _dense(a, "add"); _dense(b, "add")
    out = torch.empty_like(a)
    _call("nfdp_add_f32", _ptr(a), _ptr(b), _ptr(out), a.numel(), _stream())
    return out


def concat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat((a, b), dim=channel) for NHWC tensors, one launch."""
    bb, h, w, ca = a.shape
    cb = b.shape[-1]
    _need(a, torch.float32, "concat_channels"); _need(b, torch.float32, "concat_channels")
    if tuple(b.shape[:3]) != (bb, h, w):
        raise ValueError(f"concat_channels: shape mismatch {tuple(a.shape)} / {tuple(b.shape)}")
    out = torch.empty((bb, h, w, ca + cb), device=a.device, dtype=torch.float32)
    _call("nfdp_concat_channels_f32", _ptr(a), _ld(a, "concat_channels"), ca, _ptr(b), _ld(b, "concat_channels"), cb, _ptr(out), ca + cb,
          bb * h * w, _stream(), label="concat_channels", nbytes=8.0 * out.numel())
    return out


# ----------------------------------------------------------------------------- diffusion prior: process steps
def q_sample(x0, noise, t, sqrt_ac, sqrt_1mac):
    _dense(x0, "q_sample"); _dense(noise, "q_sample")
    out = torch.empty_like(x0)
    b = x0.shape[0]
    _call("nfdp_q_sample_f32", _ptr(x0), _ptr(noise), _ptr(t), _ptr(sqrt_ac), _ptr(sqrt_1mac), _ptr(out), b, x0.numel() // b,
          _stream())
    return out


def p_sample(x, eps_hat, noise, t, sr, srm1, c1, c2, logvar, want_x0: bool = True, clip: bool = True, out=None):
    _dense(x, "p_sample"); _dense(eps_hat, "p_sample")
    out = torch.empty_like(x) if out is None else out          # out may be x itself (elementwise, in place)
    x0 = torch.empty_like(x) if want_x0 else None
    b = x.shape[0]
    _call("nfdp_p_sample_f32", _ptr(x), _ptr(eps_hat), _ptr(noise), _ptr(t), _ptr(sr), _ptr(srm1), _ptr(c1), _ptr(c2), _ptr(logvar),
          _ptr(out), _ptr(x0), b, x.numel() // b, int(clip), _stream())
    return out, x0


def ddim_step(x, model_out, noise, sr, srm1, sqrt_an, c, sigma, last: bool, x0_coef=None, derive_noise: bool = False,
              want_x0: bool = False):
    """One DDIM step with host-side scalars (nfdp_ddim_step_ex_f32).  x0_coef = (a, b): x0 = clamp(a x - b model_out) (default
    (sr, srm1): the pred_noise objective); derive_noise: the update's noise is (sr x - x0) / srm1.  -> out, or (out, x0)."""
    out = torch.empty_like(x)
    x0 = torch.empty_like(x) if want_x0 else None
    a, b = (sr, srm1) if x0_coef is None else x0_coef
    _call("nfdp_ddim_step_ex_f32", _ptr(x), _ptr(model_out), _ptr(noise), _ptr(out), _ptr(x0), x.numel(), float(sr), float(srm1),
          float(a), float(b), int(derive_noise), float(sqrt_an), float(c), float(sigma), int(last), _stream())
    return (out, x0) if want_x0 else out


def ddim_step_table(x, model_out, noise, table, step, t_next, x0_out=None) -> None:
    """In-place DDIM step with the scalars of row step[0] of `table` ([S,8] or [S,12] fp32); t_next [B] int64 gets the next step's
    time; x0_out (optional) the clamped x0 prediction."""
    _call("nfdp_ddim_step_table_ex_f32", _ptr(x), _ptr(model_out), _ptr(noise), _ptr(x), _ptr(x0_out), x.numel(), _ptr(table),
          table.shape[1], _ptr(step), table.shape[0], _ptr(t_next), 0 if t_next is None else t_next.numel(), _stream())


def noise_from_start(x_t, x_start, t, sr, srm1) -> torch.Tensor:
    """(sqrt_recip_ac[t] x_t - x_start) / sqrt_recipm1_ac[t]  (nfdp_noise_from_start_f32)."""
    _dense(x_t, "noise_from_start"); _dense(x_start, "noise_from_start")
    out = torch.empty_like(x_t)
    b = x_t.shape[0]
    _call("nfdp_noise_from_start_f32", _ptr(x_t), _ptr(x_start), _ptr(t), _ptr(sr), _ptr(srm1), _ptr(out), b, x_t.numel() // b, _stream())
    return out


def axpb(x: torch.Tensor, a: float, b: float, add_first: bool = False) -> torch.Tensor:
    """a x + b, or (x + b) a with add_first (nfdp_axpb_f32)."""
    _need(x, torch.float32, "axpb")
    x = x.contiguous()
    out = torch.empty_like(x)
    _call("nfdp_axpb_f32", _ptr(x), float(a), float(b), _ptr(out), x.numel(), int(add_first), _stream())
    return out


def diffusion_loss(pred, target, t, p2w, l1: bool) -> torch.Tensor:
    _dense(pred, "diffusion_loss"); _dense(target, "diffusion_loss")
    b = pred.shape[0]
    ws = torch.empty(b, device=pred.device, dtype=torch.float32)
    out = torch.empty(1, device=pred.device, dtype=torch.float32)
    _call("nfdp_diffusion_loss_f32", _ptr(pred), _ptr(target), _ptr(t), _ptr(p2w), b, pred.numel() // b, int(l1), _ptr(ws), _ptr(out),
          _stream())
    return out[0]


def vb_term(x_start, x_t, eps, t, sr, srm1, c1, c2, logvar, clip: bool = True, total=None) -> torch.Tensor:
    """Per-sample variational-bound term (fp32 [B]); `total` (fp64 [B]) accumulates it in place when given."""
    for a in (x_start, x_t, eps):
        _need(a, torch.float32, "vb_term"); _dense(a, "vb_term")
    if x_t.shape != x_start.shape or eps.shape != x_start.shape:
        raise ValueError(f"vb_term: shape mismatch {tuple(x_start.shape)} / {tuple(x_t.shape)} / {tuple(eps.shape)}")
    _need(t, torch.int64, "vb_term")
    b = x_start.shape[0]
    if total is not None:
        _need(total, torch.float64, "vb_term")
    out = torch.empty(b, device=x_start.device, dtype=torch.float32)
    _call("nfdp_vb_term_f32", _ptr(x_start), _ptr(x_t), _ptr(eps), _ptr(t), _ptr(sr), _ptr(srm1), _ptr(c1), _ptr(c2), _ptr(logvar),
          b, x_start.numel() // max(b, 1), int(clip), _ptr(out), _ptr(total) if total is not None else None, _stream())
    return out


def prior_kl(x_start, sqrt_ac, log_one_minus_ac, t_index: int) -> torch.Tensor:
    _need(x_start, torch.float32, "prior_kl"); _dense(x_start, "prior_kl")
    b = x_start.shape[0]
    out = torch.empty(b, device=x_start.device, dtype=torch.float32)
    _call("nfdp_prior_kl_f32", _ptr(x_start), _ptr(sqrt_ac), _ptr(log_one_minus_ac), int(t_index), b, x_start.numel() // max(b, 1),
          _ptr(out), _stream())
    return out


# ----------------------------------------------------------------------------- backward (training step)
def _dt(t: torch.Tensor) -> int:
    return BF16 if t.dtype == torch.bfloat16 else F32


def conv_wgrad(xh: torch.Tensor, dyh: torch.Tensor, k: int, compute: int = F32, out: Optional[torch.Tensor] = None,
               accumulate: bool = False, form: int = 0, form_c: int = 0) -> torch.Tensor:
    """-> dw [Cout,Cin,k,k] fp32 of a stride-1 'same' conv; xh, dyh pixel-strided NHWC, fp32 (or bf16 with compute=BF16).
    compute=BF16 runs the bf16-MFMA kernel where its alignment rules allow (channel counts and pixel strides multiples of 4),
    else the exact-fp32 one.  out (dense fp32, e.g. the parameter's .grad) receives the result, added to its contents when
    accumulate; form 1 / 2 (k = 1, xh = im2col rows of a form_c-channel tensor) store the 3x3 weight gradient [Cout,form_c,3,3] /
    the mirrored [form_c,Cout,3,3] of the zero conv directly (nfdp_conv_wgrad_bf16_ex)."""
    for a in (xh, dyh):
        if not a.is_cuda:
            raise NfdpError("conv_wgrad: the HIP path needs GPU tensors (no CPU fallback)")
        if a.dtype not in (torch.float32, torch.bfloat16):
            raise NfdpError(f"conv_wgrad: expected fp32 or bf16, got {a.dtype}")
    b, h, w, cin = xh.shape
    cout = dyh.shape[-1]
    if tuple(dyh.shape[:3]) != (b, h, w):
        raise ValueError(f"conv_wgrad: shape mismatch {tuple(xh.shape)} / {tuple(dyh.shape)}")
    plain = out is None and form == 0
    if plain and compute == BF16 and k > 1 and cout * 4 <= cin and cin % 4 == 0 and cout % 4 == 0 and cout * k * k <= 512:
        # narrow output (ZeroConv2d 512 -> C): gather the k*k taps over the NARROW tensor instead, so the wide one is read
        # once instead of k*k times:  dW[o][i][ky][kx] = W'[i][o][k-1-ky][k-1-kx]  with  W' = wgrad(x := dy, dy := x)
        return conv_flip_weights(conv_wgrad(dyh, xh, k, compute=compute))
    if plain and compute == BF16 and cin % 4 != 0 and cout % 4 == 0 and _ld(xh, "conv_wgrad") >= (cin + 3) // 4 * 4:
        # x is a channel slice (x_a = first C/2 channels, C/2 = 6) of a wider pixel row: run the bf16 kernel on the slice
        # rounded up to a multiple of 4 channels (valid memory of the same row) and drop the extra input columns
        c4 = (cin + 3) // 4 * 4
        wide = xh.as_strided((b, h, w, c4), xh.stride())
        return conv_wgrad(wide, dyh, k, compute=compute)[:, :cin].contiguous()
    shape = {0: (cout, cin, k, k), 1: (cout, form_c, 3, 3), 2: (form_c, cout, 3, 3)}[form]
    if out is None:
        dw = torch.empty(shape, device=xh.device, dtype=torch.float32)
        accumulate = False
    else:
        _need(out, torch.float32, "conv_wgrad out"); _dense(out, "conv_wgrad out")
        if tuple(out.shape) != shape:
            raise ValueError(f"conv_wgrad: out has shape {tuple(out.shape)}, expected {shape}")
        dw = out
    if compute == BF16 and cin % 4 == 0 and cout % 4 == 0:
        xl, dl = _ld(xh, "conv_wgrad"), _ld(dyh, "conv_wgrad")
        if xl % 4 == 0 and dl % 4 == 0 and xh.data_ptr() % 16 == 0 and dyh.data_ptr() % 16 == 0:
            ws = torch.empty(lib().nfdp_conv_wgrad_bf16_ws_bytes(b, h, w, cin, cout, k), device=xh.device, dtype=torch.uint8)
            _call("nfdp_conv_wgrad_bf16_ex", _ptr(xh), _dt(xh), xl, _ptr(dyh), _dt(dyh), dl, b, h, w, cin, cout, k, _ptr(dw), _ptr(ws),
                  int(form), int(form_c), int(accumulate), _stream(), label=f"conv_wgrad_bf16 k{k} {cin}->{cout} hw{h}",
                  flops=2.0 * b * h * w * cin * cout * k * k)
            return dw
    if xh.dtype != torch.float32 or dyh.dtype != torch.float32:          # exact-fp32 kernel: fp32 operands only
        xh, dyh = xh.float(), dyh.float()
    ws = torch.empty(lib().nfdp_conv_wgrad_ws_bytes(b, h, w, cin, cout, k), device=xh.device, dtype=torch.uint8)
    _call("nfdp_conv_wgrad_f32_ex", _ptr(xh), _ld(xh, "conv_wgrad"), _ptr(dyh), _ld(dyh, "conv_wgrad"), b, h, w, cin, cout, k,
          _ptr(dw), _ptr(ws), int(form), int(form_c), int(accumulate), _stream(), label=f"conv_wgrad k{k} {cin}->{cout} hw{h}",
          flops=2.0 * b * h * w * cin * cout * k * k)
    return dw


def blocked_to_rows(t: torch.Tensor) -> torch.Tensor:
    """[..., 512] bf16 tensor in the fused kernels' blocked layout (32-pixel tiles of [hb 16][s 2][g 2][pixel 32][8]) -> ordinary
    [pixel][512] rows, same shape.  torch indexing only (tests / debugging; nothing on the product path needs it)."""
    n = t.numel() // 512
    return t.reshape(n // 32, 16, 2, 2, 32, 8).permute(0, 4, 1, 2, 3, 5).reshape(t.shape).contiguous()


def rows_to_blocked(t: torch.Tensor) -> torch.Tensor:
    n = t.numel() // 512
    return t.reshape(n // 32, 32, 16, 2, 2, 8).permute(0, 2, 3, 4, 1, 5).reshape(t.shape).contiguous()


def conv_wgrad_blocked(xh: torch.Tensor, dyh: torch.Tensor, dw: torch.Tensor, accumulate: bool, form: int = 0, form_c: int = 0,
                       x_blocked: bool = False, dy_blocked: bool = False) -> None:
    """Plain dW of a 1x1 conv on bf16 operands, either of which may be a 512-channel tensor in the blocked layout
    (nfdp_conv_wgrad_bf16_affine without the ActNorm outputs)."""
    _need(xh, torch.bfloat16, "conv_wgrad_blocked"); _need(dyh, torch.bfloat16, "conv_wgrad_blocked")
    _dense(xh, "conv_wgrad_blocked"); _dense(dyh, "conv_wgrad_blocked"); _need(dw, torch.float32, "conv_wgrad_blocked dw"); _dense(dw, "dw")
    cin, cout = xh.shape[-1], dyh.shape[-1]
    npix = xh.numel() // cin
    shape = {0: (cout, cin, 1, 1), 1: (cout, form_c, 3, 3), 2: (form_c, cout, 3, 3)}[form]
    if dyh.numel() // cout != npix or tuple(dw.shape) != shape:
        raise ValueError(f"conv_wgrad_blocked: shapes {tuple(xh.shape)} / {tuple(dyh.shape)} / dw {tuple(dw.shape)} (expected {shape})")
    ws = torch.empty(lib().nfdp_conv_wgrad_bf16_ws_bytes(1, 1, npix, cin, cout, 1), device=xh.device, dtype=torch.uint8)
    _call("nfdp_conv_wgrad_bf16_affine", _ptr(xh), cin, _ptr(dyh), cout, npix, cin, cout, _ptr(dw), _ptr(ws), int(form), int(form_c),
          int(accumulate), int(x_blocked) | (int(dy_blocked) << 1), None, None, None, None, _stream(),
          label=f"conv_wgrad_bf16_blocked {cin}->{cout}", flops=2.0 * npix * cin * cout)


def conv_wgrad_affine(xh: torch.Tensor, dyh: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, dw: torch.Tensor,
                      dscale: torch.Tensor, dbias: torch.Tensor, accumulate: bool, form: int = 0, form_c: int = 0,
                      x_blocked: bool = False, dy_blocked: bool = False) -> None:
    """dW, dscale, dbias of a Conv2dActNorm layer from ONE GEMM over pixels (nfdp_conv_wgrad_bf16_affine): xh [..., Cin] bf16 (the layer's
    input, or its im2col rows with form=1), dyh [..., Cout] bf16 = gradient at the conv's raw output; weight / bias: the layer's
    parameters; dw / dscale / dbias: dense fp32 outputs (added to when accumulate)."""
    _need(xh, torch.bfloat16, "conv_wgrad_affine"); _need(dyh, torch.bfloat16, "conv_wgrad_affine")
    _dense(xh, "conv_wgrad_affine"); _dense(dyh, "conv_wgrad_affine")
    for t in (weight, bias, dw, dscale, dbias):
        _need(t, torch.float32, "conv_wgrad_affine"); _dense(t, "conv_wgrad_affine")
    cin, cout = xh.shape[-1], dyh.shape[-1]
    npix = xh.numel() // cin
    if dyh.numel() // cout != npix:
        raise ValueError(f"conv_wgrad_affine: shape mismatch {tuple(xh.shape)} / {tuple(dyh.shape)}")
    shape = {0: (cout, cin, 1, 1), 1: (cout, form_c, 3, 3)}[form]
    if tuple(dw.shape) != shape or tuple(weight.shape) != shape or dscale.numel() != cout or dbias.numel() != cout or bias.numel() != cout:
        raise ValueError(f"conv_wgrad_affine: dw {tuple(dw.shape)} / weight {tuple(weight.shape)}, expected {shape}")
    ws = torch.empty(lib().nfdp_conv_wgrad_bf16_ws_bytes(1, 1, npix, cin, cout, 1), device=xh.device, dtype=torch.uint8)
    _call("nfdp_conv_wgrad_bf16_affine", _ptr(xh), cin, _ptr(dyh), cout, npix, cin, cout, _ptr(dw), _ptr(ws), int(form), int(form_c),
          int(accumulate), int(x_blocked) | (int(dy_blocked) << 1), _ptr(weight), _ptr(bias), _ptr(dscale), _ptr(dbias), _stream(),
          label=f"conv_wgrad_bf16_affine {cin}->{cout}", flops=2.0 * npix * cin * cout)


def coupling_wgrad3(problems, accumulate: bool = True) -> None:
    """The weight gradients of one flow's coupling network in one GEMM launch + one reduce launch (nfdp_coupling_wgrad3_bf16).
    problems: 1..3 dicts with keys x (bf16 rows [..., Cin] or a BLOCKED 512-channel tensor when x_blocked), dy (BLOCKED [..., 512] bf16),
    dw (fp32, dense), form / form_c (as conv_wgrad_affine), optional weight / bias / dscale / dbias (the layer's ActNorm: all or none),
    x_blocked."""
    n = len(problems)
    if not 1 <= n <= 3:
        raise ValueError("coupling_wgrad3: 1..3 problems")
    npix = None
    for pr in problems:
        xh, dyh = pr["x"], pr["dy"]
        _need(xh, torch.bfloat16, "coupling_wgrad3"); _need(dyh, torch.bfloat16, "coupling_wgrad3")
        _dense(xh, "coupling_wgrad3"); _dense(dyh, "coupling_wgrad3")
        if dyh.shape[-1] != 512:
            raise ValueError("coupling_wgrad3: dy must have 512 channels")
        cin = xh.shape[-1]
        if npix is None:
            npix = dyh.numel() // 512
        if dyh.numel() // 512 != npix or xh.numel() // cin != npix:
            raise ValueError(f"coupling_wgrad3: pixel counts differ ({tuple(xh.shape)} / {tuple(dyh.shape)})")
        form, form_c = int(pr.get("form", 0)), int(pr.get("form_c", 0))
        shape = {0: (512, cin, 1, 1), 1: (512, form_c, 3, 3), 2: (form_c, 512, 3, 3)}[form]
        aff = [pr.get(k) for k in ("weight", "bias", "dscale", "dbias")]
        if any(t is None for t in aff) != all(t is None for t in aff):
            raise ValueError("coupling_wgrad3: weight / bias / dscale / dbias go together")
        for t in [pr["dw"]] + [t for t in aff if t is not None]:
            _need(t, torch.float32, "coupling_wgrad3"); _dense(t, "coupling_wgrad3")
        if tuple(pr["dw"].shape) != shape or (aff[0] is not None and tuple(aff[0].shape) != shape):
            raise ValueError(f"coupling_wgrad3: dw {tuple(pr['dw'].shape)}, expected {shape}")
    ints = lambda vs: (ctypes.c_int * n)(*[int(v) for v in vs])
    tab = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else _ptr(t) for t in ts])
    cins = ints(pr["x"].shape[-1] for pr in problems)
    nbytes = lib().nfdp_coupling_wgrad3_ws_bytes(npix, cins, n)
    if nbytes == 0:
        raise NfdpError(f"coupling_wgrad3: unsupported shapes (npix={npix}, cin={[pr['x'].shape[-1] for pr in problems]})")
    ws = torch.empty(nbytes, device=problems[0]["x"].device, dtype=torch.uint8)
    _call("nfdp_coupling_wgrad3_bf16", tab(pr["x"] for pr in problems), ints(pr["x"].shape[-1] for pr in problems),
          ints(bool(pr.get("x_blocked", False)) for pr in problems), tab(pr["dy"] for pr in problems), cins,
          tab(pr["dw"] for pr in problems), ints(pr.get("form", 0) for pr in problems), ints(pr.get("form_c", 0) for pr in problems),
          tab(pr.get("weight") for pr in problems), tab(pr.get("bias") for pr in problems), tab(pr.get("dscale") for pr in problems),
          tab(pr.get("dbias") for pr in problems), n, npix, _ptr(ws), int(accumulate), _stream(),
          label=f"coupling_wgrad3 x{n}", flops=sum(2.0 * npix * 512 * pr["x"].shape[-1] for pr in problems))


def coupling_wgrad3_batch(flows, accumulate: bool = True) -> None:
    """coupling_wgrad3 for several flows of a level in ONE GEMM launch + ONE reduce launch (nfdp_coupling_wgrad3_batch_bf16): `flows` is a
    list of problem lists as coupling_wgrad3 takes them, all of the same shapes / forms.  More than the library's 16 flows per launch are
    sent in chunks.  A flow's result does not depend on how many flows travel with it."""
    if not flows:
        return
    n = len(flows[0])
    if not 1 <= n <= 3 or any(len(pr) != n for pr in flows):
        raise ValueError("coupling_wgrad3_batch: 1..3 problems per flow, the same number for every flow")
    first = flows[0]
    npix = first[0]["dy"].numel() // 512
    for prs in flows:
        for q, pr in enumerate(prs):
            xh, dyh, ref = pr["x"], pr["dy"], first[q]
            _need(xh, torch.bfloat16, "coupling_wgrad3_batch"); _need(dyh, torch.bfloat16, "coupling_wgrad3_batch")
            _dense(xh, "coupling_wgrad3_batch"); _dense(dyh, "coupling_wgrad3_batch")
            cin = xh.shape[-1]
            if dyh.shape[-1] != 512 or dyh.numel() // 512 != npix or xh.numel() // cin != npix or cin != ref["x"].shape[-1]:
                raise ValueError(f"coupling_wgrad3_batch: shapes differ between flows ({tuple(xh.shape)} / {tuple(dyh.shape)})")
            for k in ("form", "form_c", "x_blocked"):
                if pr.get(k, 0) != ref.get(k, 0):
                    raise ValueError(f"coupling_wgrad3_batch: '{k}' differs between flows")
            form, form_c = int(pr.get("form", 0)), int(pr.get("form_c", 0))
            shape = {0: (512, cin, 1, 1), 1: (512, form_c, 3, 3), 2: (form_c, 512, 3, 3)}[form]
            aff = [pr.get(k) for k in ("weight", "bias", "dscale", "dbias")]
            if any(t is None for t in aff) != all(t is None for t in aff):
                raise ValueError("coupling_wgrad3_batch: weight / bias / dscale / dbias go together")
            for t in [pr["dw"]] + [t for t in aff if t is not None]:
                _need(t, torch.float32, "coupling_wgrad3_batch"); _dense(t, "coupling_wgrad3_batch")
            if tuple(pr["dw"].shape) != shape or (aff[0] is not None and tuple(aff[0].shape) != shape):
                raise ValueError(f"coupling_wgrad3_batch: dw {tuple(pr['dw'].shape)}, expected {shape}")
    ints = lambda vs: (ctypes.c_int * n)(*[int(v) for v in vs])
    cins = ints(pr["x"].shape[-1] for pr in first)
    per_flow = lib().nfdp_coupling_wgrad3_batch_ws_bytes(npix, cins, n)
    if per_flow == 0:
        raise NfdpError(f"coupling_wgrad3_batch: unsupported shapes (npix={npix}, cin={[pr['x'].shape[-1] for pr in first]})")
    cap = int(lib().nfdp_coupling_wgrad3_batch_max_flows())
    for c0 in range(0, len(flows), cap):
        chunk = flows[c0:c0 + cap]
        nf = len(chunk)
        tab = lambda key: (ctypes.c_void_p * (nf * n))(*[None if pr.get(key) is None else _ptr(pr[key]) for prs in chunk for pr in prs])
        ws = torch.empty(per_flow * nf, device=first[0]["x"].device, dtype=torch.uint8)
        _call("nfdp_coupling_wgrad3_batch_bf16", tab("x"), ints(pr["x"].shape[-1] for pr in first),
              ints(bool(pr.get("x_blocked", False)) for pr in first), tab("dy"), cins, tab("dw"), ints(pr.get("form", 0) for pr in first),
              ints(pr.get("form_c", 0) for pr in first), tab("weight"), tab("bias"), tab("dscale"), tab("dbias"), n, nf, npix, _ptr(ws),
              int(accumulate), _stream(), label=f"coupling_wgrad3_batch x{n} x{nf} flows",
              flops=nf * sum(2.0 * npix * 512 * pr["x"].shape[-1] for pr in first))


def conv_flip_weights(w_oihw: torch.Tensor) -> torch.Tensor:
    _need(w_oihw, torch.float32, "conv_flip_weights"); _dense(w_oihw, "conv_flip_weights")
    o, i, k, _ = w_oihw.shape
    wt = torch.empty((i, o, k, k), device=w_oihw.device, dtype=torch.float32)
    _call("nfdp_conv_flip_weights_f32", _ptr(w_oihw), _ptr(wt), o, i, k, _stream())
    return wt


def scale_bwd(dyh, yh=None, mul=None, f: float = 1.0, relu: bool = False, want_dc: bool = True, dc_dtype=None,
              grads_out=None):
    """Backward of y = act((u + add) * exp(f*mul)) -> (dc or None, dmul [C] or None, dadd [C]).  dyh / yh fp32 or bf16;
    dc_dtype (torch dtype, default: dyh's... fp32 unless yh is bf16) selects the storage of dc.  grads_out = (dmul_acc, dadd_acc):
    dense fp32 [C]-sized tensors (the parameters' .grad) the two sums are ADDED to in the finishing launch."""
    if dyh.dtype not in (torch.float32, torch.bfloat16) or not dyh.is_cuda:
        raise NfdpError("scale_bwd: expected an fp32 or bf16 GPU tensor")
    c = dyh.shape[-1]
    npix = dyh.numel() // c
    dev = dyh.device
    if dc_dtype is None:
        dc_dtype = torch.bfloat16 if (yh is not None and yh.dtype == torch.bfloat16) else torch.float32
    dc = torch.empty(dyh.shape, device=dev, dtype=dc_dtype) if want_dc else None
    if grads_out is not None:
        dmul, dadd = grads_out
        for g in (dmul, dadd):
            _need(g, torch.float32, "scale_bwd grads_out"); _dense(g, "scale_bwd grads_out")
            if g.numel() != c:
                raise ValueError(f"scale_bwd: grads_out tensors must hold {c} elements")
    else:
        dmul = torch.empty(c, device=dev, dtype=torch.float32) if yh is not None else None
        dadd = torch.empty(c, device=dev, dtype=torch.float32)
    ws = torch.empty(lib().nfdp_scale_bwd_ws_bytes(npix, c), device=dev, dtype=torch.uint8)
    _call("nfdp_scale_bwd_ex", _ptr(dyh), _dt(dyh), _ld4(dyh), _ptr(yh), _dt(yh) if yh is not None else F32,
          _ld4(yh) if yh is not None else c, _ptr(mul), float(f), int(relu), npix, c, _ptr(dc), _dt(dc) if dc is not None else F32, c,
          _ptr(dmul), _ptr(dadd), _ptr(ws), int(grads_out is not None), _stream())
    return dc, dmul, dadd


def _ld4(t: torch.Tensor) -> int:
    """Pixel stride of a [..., C] tensor of any rank >= 2 that is dense or a channel slice of a dense tensor."""
    if t.dim() == 4:
        return _ld(t, "backward")
    if t.stride(-1) != 1 and t.shape[-1] > 1:
        raise NfdpError("backward: channel stride must be 1")
    return t.stride(-2) if t.dim() >= 2 and t.shape[-2] > 1 else t.shape[-1]


def coupling_bwd(xh, hh, dyh, gl: Optional[torch.Tensor]):
    for a in (xh, hh, dyh):
        _need(a, torch.float32, "coupling_bwd"); _dense(a, "coupling_bwd")
    b, h, w, c = xh.shape
    dx, dh = torch.empty_like(xh), torch.empty_like(hh)
    if gl is not None:
        _need(gl, torch.float64, "coupling_bwd")
    _call("nfdp_coupling_bwd_f32", _ptr(xh), _ptr(hh), _ptr(dyh), _ptr(gl), b, h * w, c, _ptr(dx), _ptr(dh), _stream())
    return dx, dh


def flow_link_bwd(dx_in: torch.Tensor, part: torch.Tensor, *, head=None, coupling=None, want_a: bool = True):
    """One launch between two reverse coupling networks of the backward pass (nfdp_flow_link_bwd_f32).  dx_in [B,H,W,C] fp32;
    head = (xpre, w [C,C] = the 1x1 weight, scale [C], bias [C]) of flow i; coupling = (ypost, hout, gl fp64 [B] or None, logs3 [C], f3) of
    flow i-1; part: this call's fp64 partial buffer (flow_link_part).  -> (dx_out, dc3 or None, a or None)."""
    _need(dx_in, torch.float32, "flow_link_bwd"); _dense(dx_in, "flow_link_bwd")
    b, h, w, c = dx_in.shape
    npix = b * h * w
    dx_out = torch.empty_like(dx_in)
    xpre = wt = scale = bias = a = None
    if head is not None:
        xpre, wt, scale, bias = head
        for t in (xpre, wt, scale, bias):
            _need(t, torch.float32, "flow_link_bwd head"); _dense(t, "flow_link_bwd head")
        a = torch.empty_like(dx_in) if want_a else None
    ypost = hout = gl = logs3 = dc3 = None
    f3 = 0.0
    if coupling is not None:
        ypost, hout, gl, logs3, f3 = coupling
        for t in (ypost, hout, logs3):
            _need(t, torch.float32, "flow_link_bwd coupling"); _dense(t, "flow_link_bwd coupling")
        if gl is not None:
            _need(gl, torch.float64, "flow_link_bwd gl")
        dc3 = torch.empty_like(dx_in)
    _call("nfdp_flow_link_bwd_f32", _ptr(dx_in), _ptr(xpre), _ptr(wt), _ptr(scale), _ptr(bias), _ptr(a), _ptr(ypost), _ptr(hout), _ptr(gl),
          _ptr(logs3), float(f3), _ptr(dx_out), _ptr(dc3), _ptr(part), npix, h * w, c, _stream(), label=f"flow_link_bwd<{c}>",
          nbytes=4.0 * npix * c * 6)
    return dx_out, dc3, a


def flow_link_part(n_calls: int, npix: int, c: int, device) -> torch.Tensor:
    """Partial buffer for n_calls flow_link_bwd launches of a level: per call [blocks, 4, C] fp64 + [blocks, C*C] fp32."""
    per = lib().nfdp_flow_link_bwd_part_bytes(npix, c) // 8
    return torch.empty((n_calls, per), device=device, dtype=torch.float64)


def flow_link_finish(part: torch.Tensor, npix: int, c: int, f3: float, dscale, dbias, dlogs3, dbias3, accumulate: bool, dw=None) -> None:
    """Reduce the partials of all link calls of a level in one launch; the four lists hold one fp32 [C] tensor or None per call, dw one
    dense fp32 tensor of C*C elements (the 1x1 conv's weight gradient of the call's head) or None."""
    n = part.shape[0]
    tab = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    if dw is None:
        dw = [None] * n
    if len(dw) != n:
        raise ValueError("flow_link_finish: one entry per call")
    for t in dw:
        if t is not None:
            _need(t, torch.float32, "flow_link_finish"); _dense(t, "flow_link_finish")
            if t.numel() != c * c:
                raise ValueError("flow_link_finish: dw entries are [C, C]")
    for ts in (dscale, dbias, dlogs3, dbias3):
        if len(ts) != n:
            raise ValueError("flow_link_finish: one entry per call")
        for t in ts:
            if t is not None:
                _need(t, torch.float32, "flow_link_finish"); _dense(t, "flow_link_finish")
                if t.numel() != c:
                    raise ValueError("flow_link_finish: outputs are [C]")
    _call("nfdp_flow_link_finish_f32", _ptr(part), n, npix, c, float(f3), tab(dscale), tab(dbias), tab(dlogs3), tab(dbias3), tab(dw),
          int(accumulate), _stream(), label="flow_link_finish")


def gauss_logp_bwd(zh, h, h_mode: int, gl: torch.Tensor):
    _need(gl, torch.float64, "gauss_logp_bwd")
    b, hh, ww, cz = zh.shape
    dz = torch.empty((b, hh, ww, cz), device=zh.device, dtype=torch.float32)
    dh = torch.empty((b, hh, ww, 2 * cz), device=zh.device, dtype=torch.float32)
    _call("nfdp_gauss_logp_bwd_f32", _ptr(zh), _ld(zh, "gauss_logp_bwd"), _ptr(h), h_mode, _ptr(gl), b, hh * ww, cz, _ptr(dz),
          _ptr(dh), _stream())
    return dz, dh


# ----------------------------------------------------------------------------- backward of the U-Net glue ops
def groupnorm_silu_bwd(xh, dyh, groups: int, gamma, beta, scale_shift, eps: float):
    """-> (dx, dgamma [C], dbeta [C], dscale_shift [B,2C] or None)."""
    _need(xh, torch.float32, "groupnorm_silu_bwd"); _dense(xh, "groupnorm_silu_bwd"); _dense(dyh, "groupnorm_silu_bwd")
    b, h, w, c = xh.shape
    dev = xh.device
    dx = torch.empty_like(xh)
    dg_b = torch.empty((b, c), device=dev, dtype=torch.float32)
    db_b = torch.empty((b, c), device=dev, dtype=torch.float32)
    dss = torch.empty((b, 2 * c), device=dev, dtype=torch.float32) if scale_shift is not None else None
    ss_ld = 0
    if scale_shift is not None:
        if scale_shift.shape != (b, 2 * c) or scale_shift.stride(1) != 1:
            raise ValueError("groupnorm_silu_bwd: scale_shift must be a [B, 2C] row-strided view")
        ss_ld = scale_shift.stride(0)
    _call("nfdp_groupnorm_silu_bwd_f32", _ptr(xh), _ptr(dyh), b, h * w, c, groups, _ptr(gamma), _ptr(beta), _ptr(scale_shift), ss_ld,
          float(eps), _ptr(dx), _ptr(dg_b), _ptr(db_b), _ptr(dss), _stream())
    dgamma = scale_bwd(dg_b, want_dc=False)[2]                   # sum over the batch (fixed order)
    dbeta = scale_bwd(db_b, want_dc=False)[2]
    return dx, dgamma, dbeta, dss


def chan_layernorm_bwd(xh, dyh, gain, eps: float = 1e-5):
    """-> (dx, dgain [C])."""
    _need(xh, torch.float32, "chan_layernorm_bwd"); _dense(xh, "chan_layernorm_bwd"); _dense(dyh, "chan_layernorm_bwd")
    c = xh.shape[-1]
    dx, dgx = torch.empty_like(xh), torch.empty_like(xh)
    _call("nfdp_chan_layernorm_bwd_f32", _ptr(xh), _ptr(dyh), xh.numel() // c, c, _ptr(gain), float(eps), _ptr(dx), _ptr(dgx), _stream())
    return dx, scale_bwd(dgx, want_dc=False)[2]


def _attn_bwd(name, qkv, dout, heads, dim_head, scale):
    _need(qkv, torch.float32, name); _dense(qkv, name); _dense(dout, name)
    b, h, w, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    _call(name, _ptr(qkv), _ptr(dout), _ptr(dqkv), b, h * w, heads, dim_head, float(scale), _stream())
    return dqkv


def linear_attention_bwd(qkv, dout, heads: int, dim_head: int, scale: float):
    return _attn_bwd("nfdp_linear_attention_bwd_f32", qkv, dout, heads, dim_head, scale)


def attention_bwd(qkv, dout, heads: int, dim_head: int, scale: float):
    return _attn_bwd("nfdp_attention_bwd_f32", qkv, dout, heads, dim_head, scale)


def upsample2x_bwd(dyh):
    _need(dyh, torch.float32, "upsample2x_bwd"); _dense(dyh, "upsample2x_bwd")
    b, h2, w2, c = dyh.shape
    dx = torch.empty((b, h2 // 2, w2 // 2, c), device=dyh.device, dtype=torch.float32)
    _call("nfdp_upsample_nearest2x_bwd_f32", _ptr(dyh), _ptr(dx), b, c, h2 // 2, w2 // 2, _stream())
    return dx


GELU, SILU = 1, 2


def act_fwd(x, act: int):
    _need(x, torch.float32, "act_fwd"); _dense(x, "act_fwd")
    y = torch.empty_like(x)
    _call("nfdp_act_fwd_f32", _ptr(x), _ptr(y), x.numel(), act, _stream())
    return y


def act_bwd(x, dy, act: int):
    _need(x, torch.float32, "act_bwd"); _dense(x, "act_bwd"); _dense(dy, "act_bwd")
    dx = torch.empty_like(x)
    _call("nfdp_act_bwd_f32", _ptr(x), _ptr(dy), _ptr(dx), x.numel(), act, _stream())
    return dx


def diffusion_loss_bwd(pred, target, t, p2w, grad_out, l1: bool):
    _dense(pred, "diffusion_loss_bwd"); _dense(target, "diffusion_loss_bwd")
    b = pred.shape[0]
    dp = torch.empty_like(pred)
    _call("nfdp_diffusion_loss_bwd_f32", _ptr(pred), _ptr(target), _ptr(t), _ptr(p2w), _ptr(grad_out), b, pred.numel() // b, int(l1),
          _ptr(dp), _stream())
    return dp


def conv_pack_ex(w_oihw: torch.Tensor, compute: int, flip: bool = False, taps: bool = False) -> torch.Tensor:
    """conv_pack of a derived weight in one launch from the parameter: flip = the data-gradient weight (flipped taps, channels
    swapped), taps = the pointwise [tap*Co + co][ci] form of conv3x3_narrow (nfdp_conv_pack_weights_ex)."""
    _need(w_oihw, torch.float32, "conv_pack_ex"); _dense(w_oihw, "conv_pack_ex")
    co, ci, kh, kw = w_oihw.shape
    nbytes = lib().nfdp_conv_packed_bytes_ex(co, ci, kh, kw, compute, int(flip), int(taps))
    packed = torch.empty(nbytes, device=w_oihw.device, dtype=torch.uint8)
    _call("nfdp_conv_pack_weights_ex", _ptr(w_oihw), _ptr(packed), co, ci, kh, kw, compute, int(flip), int(taps), _stream())
    return packed


class PackBatch:
    """Kernel-layout copies of many conv weights refreshed by ONE launch (nfdp_conv_pack_weights_batch).  specs: [(weight OIHW fp32
    parameter, flip, taps)]; .views[i] is the packed tensor of specs[i] (same bytes as conv_pack_ex), valid after run().  The views
    live in one buffer and the job table on the device, so a captured training step replays the launch unchanged; matches() tells
    whether the parameters still live where the table points."""

    def __init__(self, specs, compute: int):
        import numpy as np
        if not specs:
            raise ValueError("PackBatch: no weights")
        self.compute = compute
        dev = specs[0][0].device
        sizes, offs, total = [], [], 0
        for w, flip, taps in specs:
            _need(w, torch.float32, "PackBatch"); _dense(w, "PackBatch")
            co, ci, kh, kw = w.shape
            n = lib().nfdp_conv_packed_bytes_ex(co, ci, kh, kw, compute, int(flip), int(taps))
            sizes.append(n); offs.append(total)
            total += -(-n // 256) * 256
        self.buf = torch.empty(total, device=dev, dtype=torch.uint8)
        self.views = [self.buf[o:o + n] for o, n in zip(offs, sizes)]
        jobs = np.zeros((len(specs), 12), dtype=np.int64)
        for i, ((w, flip, taps), v) in enumerate(zip(specs, self.views)):
            co, ci, kh, kw = w.shape
            _check(lib().nfdp_conv_pack_job(_ptr(w), _ptr(v), co, ci, kh, kw, int(flip), int(taps), jobs[i].ctypes.data))
        self.max_total = int(jobs[:, 10].max())
        self.jobs = torch.from_numpy(jobs).to(dev)
        self._key = tuple(w.data_ptr() for w, _, _ in specs)
        self._weights = [w for w, _, _ in specs]

    def matches(self) -> bool:
        return self._key == tuple(w.data_ptr() for w in self._weights)

    def run(self) -> None:
        _call("nfdp_conv_pack_weights_batch", _ptr(self.jobs), self.jobs.shape[0], self.max_total, self.compute, _stream(),
              label=f"pack_weights_batch x{self.jobs.shape[0]}")


def pack_taps(w_oihw: torch.Tensor, compute: int) -> torch.Tensor:
    """Packed weight of the pointwise GEMM of conv3x3_narrow: rows [tap*Co + co], columns ci."""
    return conv_pack_ex(w_oihw.detach().contiguous(), compute, taps=True)


def conv3x3_narrow(xh: torch.Tensor, w_oihw: torch.Tensor, *, add=None, mul=None, mul_exp: float = 0.0, relu: bool = False,
                   compute: int = F32, packed_taps: Optional[torch.Tensor] = None, co: Optional[int] = None,
                   return_taps: bool = False) -> torch.Tensor:
    """3x3 'same' conv with Cin >> Cout as a pointwise GEMM to the 9*Cout tap contributions (one wide, fast 1x1 conv)
    followed by the tap shift-sum with the conv kernel's epilogue act((acc + add) * exp(mul_exp * mul)).
    return_taps: stop after the GEMM and return T [B,H,W,9*Cout] (for tap_coupling)."""
    if w_oihw is not None:
        co, ci, k, _ = w_oihw.shape
        if k != 3:
            raise ValueError("conv3x3_narrow: 3x3 only")
    b, h, w, _ = xh.shape
    if packed_taps is None:
        packed_taps = pack_taps(w_oihw, compute)
    t = conv2d(xh, packed_taps, 9 * co, 1, compute=compute, out_dtype=F32)
    if return_taps:
        return t
    out = torch.empty((b, h, w, co), device=xh.device, dtype=torch.float32)
    _call("nfdp_tap_shift_sum_f32", _ptr(t), _ptr(out), b, h, w, co, _ptr(add), _ptr(mul), float(mul_exp), int(relu), _stream())
    return out


def pointwise_dgrad_masked_supported(npix: int, cin: int, cout: int) -> bool:
    return bool(lib().nfdp_pointwise_dgrad_masked_supported(npix, cin, cout))


def pointwise_dgrad_masked(xh: torch.Tensor, packed: torch.Tensor, y_saved: torch.Tensor, mul: torch.Tensor, f: float, grads_out=None):
    """Data gradient of a 1x1 conv (xh [B,H,W,Cin] bf16 times the packed data-gradient weight) fused with the scale/ReLU backward of
    the layer whose saved output is y_saved [B,H,W,Cout] bf16: -> (dc bf16 [B,H,W,Cout], dmul [Cout], dadd [Cout]).
    grads_out = (dmul_acc, dadd_acc): the sums are added to these dense fp32 tensors (the parameters' .grad) instead."""
    _need(xh, torch.bfloat16, "pointwise_dgrad_masked"); _need(y_saved, torch.bfloat16, "pointwise_dgrad_masked")
    b, h, w, cin = xh.shape
    cout = y_saved.shape[-1]
    dc = torch.empty((b, h, w, cout), device=xh.device, dtype=torch.bfloat16)
    if grads_out is not None:
        dmul, dadd = grads_out
        for g in (dmul, dadd):
            _need(g, torch.float32, "pointwise_dgrad_masked grads_out"); _dense(g, "pointwise_dgrad_masked grads_out")
            if g.numel() != cout:
                raise ValueError(f"pointwise_dgrad_masked: grads_out tensors must hold {cout} elements")
    else:
        dmul = torch.empty(cout, device=xh.device, dtype=torch.float32)
        dadd = torch.empty(cout, device=xh.device, dtype=torch.float32)
    ws = torch.empty(lib().nfdp_pointwise_dgrad_masked_ws_bytes(b * h * w, cout), device=xh.device, dtype=torch.uint8)
    _call("nfdp_pointwise_dgrad_masked_bf16_ex", _ptr(xh), _ld(xh, "pointwise_dgrad_masked"), _ptr(packed), _ptr(y_saved),
          _ld(y_saved, "pointwise_dgrad_masked"), _ptr(mul), float(f), b * h * w, cin, cout, _ptr(dc), cout, _ptr(dmul), _ptr(dadd),
          _ptr(ws), int(grads_out is not None), _stream(), label=f"pw_gemm_bf16<masked> k1 {cin}->{cout} hw{h}",
          flops=2.0 * b * h * w * cin * cout)
    return dc, dmul, dadd


MLP_GATHER = True      # coupling_net_tiled: gather the first conv's operands inside the coupling-network kernel (False: explicit im2col)


_MLP_FORM = 0          # host-side default of the per-call `form` argument of the tiled coupling-network entries (include/nfdp.h)


def set_mlp_rows(mode: str) -> None:
    """Kernel form of the tiled coupling networks: "auto" (the row-split form - 32-pixel workgroups, the four waves split the weights -
    where the 128-pixel form would leave most CUs idle: <= 16384 pixels), "never", "always".  bf16 results are bit-identical.
    A default of this binding, handed to the library per call (`form`)."""
    global _MLP_FORM
    _MLP_FORM = (_MLP_FORM & ~3) | {"auto": 0, "never": 1, "always": 2}[mode]


def coupling_mlp_supported(c: int, n_features: int) -> bool:
    return n_features == 512 and c <= 48 and lib().nfdp_coupling_fused_packed_bytes(c) != 0


def coupling_mlp_image_kind(xh_a: torch.Tensor, c: int, split: bool = False) -> int:
    """Which image pair coupling_net_tiled wants for this call: 2 = the 16x16x32-form images (fused_packed(2), low pieces
    fused_packed(4); C = 4, 12 beyond the row-split kernel's sizes), 1 = fused_packed(1) / fused_packed(3)."""
    b, h, w, _ = xh_a.shape
    if xh_a.dtype != torch.float32 or xh_a.data_ptr() % 8 or _ld(xh_a, "coupling_mlp_image_kind") % 2 or not MLP_GATHER:
        return 1
    return int(lib().nfdp_coupling_mlp_image_kind(c, b * h * w, int(split), _MLP_FORM))


def set_mlp_k16(on: bool) -> None:
    """Tiled coupling networks of C = 4, 12 on v_mfma_f32_16x16x32_bf16 (default) or on the 32x32x16 kernels (per-call `form` bit
    NFDP_MLP_NO_K16)."""
    global _MLP_FORM
    _MLP_FORM = (_MLP_FORM & ~4) | (0 if on else 4)


def coupling_net_tiled(xh_a: torch.Tensor, packed: torch.Tensor, c: int, bias: torch.Tensor, logs: torch.Tensor,
                       mul_exp: float, return_taps: bool = False, packed_lo: Optional[torch.Tensor] = None, kind: int = 1) -> torch.Tensor:
    """NN(x_a) of an affine coupling for any image size (bf16 mode): im2col of the C/2-channel half -> the per-pixel three-GEMM
    chain with the hidden activations on chip (nfdp_coupling_mlp_bf16) -> tap shift-sum with the ZeroConv2d epilogue.
    xh_a: [B,H,W,C/2] fp32 pixel-strided view; returns h = (log_s | t) [B,H,W,C] fp32."""
    b, h, w, _ = xh_a.shape
    t = torch.empty((b, h, w, 9 * c), device=xh_a.device, dtype=torch.float32)
    fl = 2.0 * b * h * w * (9 * (c // 2) * 512 + 512 * 512 + 9 * 512 * c)
    vw = 4 if (c // 2) % 4 == 0 else 2
    if kind == 2:
        # 16x16x32 form: packed = the fold-2 image, packed_lo = None or the fold-4 image (coupling_mlp_image_kind chose them)
        _need(xh_a, torch.float32, "coupling_net_tiled")
        _call("nfdp_coupling_mlp_gather_k16", _ptr(xh_a), _ld(xh_a, "coupling_net_tiled"), b, h, w, _ptr(packed), _ptr(packed_lo), _ptr(t),
              c, _stream(), label=f"coupling_mlp{'_x3' if packed_lo is not None else ''}<{c}> hw{h}", flops=fl)
    elif packed_lo is not None:
        # split-product form (compute "f32", set_f32_coupling("bf16x3")): packed = high-piece image, packed_lo = low-piece image
        _need(xh_a, torch.float32, "coupling_net_tiled")
        _call("nfdp_coupling_mlp_gather_bf16x3", _ptr(xh_a), _ld(xh_a, "coupling_net_tiled"), b, h, w, _ptr(packed), _ptr(packed_lo), _ptr(t),
              c, _MLP_FORM, _stream(), label=f"coupling_mlp_x3<{c}> hw{h}", flops=fl)
    elif MLP_GATHER and xh_a.dtype == torch.float32 and xh_a.data_ptr() % (4 * vw) == 0 and _ld(xh_a, "coupling_net_tiled") % vw == 0:
        # the first conv's im2col inside the kernel (no xcol matrix, no im2col launch)
        _call("nfdp_coupling_mlp_gather_bf16", _ptr(xh_a), _ld(xh_a, "coupling_net_tiled"), b, h, w, _ptr(packed), _ptr(t), c, _MLP_FORM, _stream(),
              label=f"coupling_mlp<{c}> hw{h}", flops=fl)
    else:
        xcol = im2col3x3(xh_a)
        _call("nfdp_coupling_mlp_bf16", _ptr(xcol), xcol.shape[-1], b * h * w, _ptr(packed), _ptr(t), c, _stream(),
              label=f"coupling_mlp<{c}> hw{h}", flops=fl)
    if return_taps:
        return t
    out = torch.empty((b, h, w, c), device=xh_a.device, dtype=torch.float32)
    _call("nfdp_tap_shift_sum_f32", _ptr(t), _ptr(out), b, h, w, c, _ptr(bias), _ptr(logs), float(mul_exp), 0, _stream())
    return out


_F32_FUSED = True


def set_f32_fused(on: bool) -> None:
    """compute="f32": run the coupling networks on the exact-fp32 on-chip kernel (nfdp_coupling_mlp_f32, default) or conv by conv."""
    global _F32_FUSED
    _F32_FUSED = bool(on)


_F32_COUPLING = "mfma_f32"


def set_f32_coupling(mode: str) -> None:
    """compute="f32": how the coupling networks form their fp32 products.  "mfma_f32" (default): v_mfma_f32_32x32x2_f32, exact fp32
    products.  "bf16x3": each product as hi*hi + hi*lo + lo*hi of two-piece bf16 splits of both operands (three bf16 MFMAs, fp32
    accumulation; relative product error ~4e-6, pixels of a 48-flow inverse 4-6e-5 from fp64) where the geometry has that kernel
    (C <= 24 and C = 48; wider levels stay on the fp32 matrix cores)."""
    global _F32_COUPLING
    if mode not in ("mfma_f32", "bf16x3"):
        raise ValueError(f"set_f32_coupling: {mode!r}")
    _F32_COUPLING = mode


def get_f32_coupling() -> str:
    return _F32_COUPLING


def coupling_mlp_x3_supported(xh_a: torch.Tensor, c: int, n_features: int) -> bool:
    """The split-product kernels take this coupling (mode on, C <= 24 or C = 48, pieces gathered with aligned vector loads)."""
    if _F32_COUPLING != "bf16x3" or not _F32_FUSED or n_features != 512 or (c > 24 and c != 48) or c % 4 or xh_a.dtype != torch.float32:
        return False
    if lib().nfdp_coupling_fused_packed_bytes(c) == 0:
        return False
    vw = 4 if (c // 2) % 4 == 0 else 2
    return xh_a.data_ptr() % (4 * vw) == 0 and _ld(xh_a, "coupling_mlp_x3") % vw == 0


def coupling_mlp_f32_supported(c: int, n_features: int) -> bool:
    return _F32_FUSED and n_features == 512 and lib().nfdp_coupling_mlp_f32_packed_bytes(c) != 0


def coupling_mlp_f32_pack(w1, scale1, bias1, w2, scale2, bias2, w3, out=None) -> torch.Tensor:
    c = w3.shape[0]
    n = lib().nfdp_coupling_mlp_f32_packed_bytes(c)
    if n == 0:
        raise NfdpError(f"coupling_mlp_f32_pack: C={c} unsupported")
    if out is None or out.numel() != n:
        out = torch.empty(n, device=w1.device, dtype=torch.uint8)
    for t in (w1, scale1, bias1, w2, scale2, bias2, w3):
        _need(t, torch.float32, "coupling_mlp_f32_pack"); _dense(t, "coupling_mlp_f32_pack")
    _call("nfdp_coupling_mlp_f32_pack", _ptr(w1), _ptr(scale1), _ptr(bias1), _ptr(w2), _ptr(scale2), _ptr(bias2), _ptr(w3), _ptr(out), c,
          _stream())
    return out


def coupling_net_tiled_f32(xh_a: torch.Tensor, packed: torch.Tensor, c: int, bias: torch.Tensor, logs: torch.Tensor,
                           mul_exp: float, variant: int = 0) -> torch.Tensor:
    """NN(x_a) of an affine coupling in exact fp32, hidden activations on chip: nfdp_coupling_mlp_f32 (3x3 gather + three fp32-MFMA
    GEMMs) -> tap shift-sum with the ZeroConv2d epilogue.  xh_a: [B,H,W,C/2] fp32 pixel-strided view; returns (log_s | t) [B,H,W,C]."""
    _need(xh_a, torch.float32, "coupling_net_tiled_f32")
    b, h, w, _ = xh_a.shape
    t = torch.empty((b, h, w, 9 * c), device=xh_a.device, dtype=torch.float32)
    _call("nfdp_coupling_mlp_f32_ex", _ptr(xh_a), _ld(xh_a, "coupling_net_tiled_f32"), b, h, w, _ptr(packed), _ptr(t), c, int(variant),
          _stream(), label=f"coupling_mlp_f32<{c}> hw{h}", flops=2.0 * b * h * w * (9 * (c // 2) * 512 + 512 * 512 + 9 * 512 * c))
    out = torch.empty((b, h, w, c), device=xh_a.device, dtype=torch.float32)
    _call("nfdp_tap_shift_sum_f32", _ptr(t), _ptr(out), b, h, w, c, _ptr(bias), _ptr(logs), float(mul_exp), 0, _stream())
    return out


def use_narrow_conv(cin: int, cout: int, k: int) -> bool:
    return k == 3 and cout <= 64 and cin >= 4 * cout and cin >= 128


def use_tap_conv(cin: int, cout: int, k: int, compute: int) -> bool:
    """Inference form of the 512 -> C zero conv in bf16 mode (any C up to 192: the deeper levels of 4- and 5-level models)."""
    return compute == BF16 and k == 3 and cout <= 192 and cin >= 2 * cout and cin >= 128 and cin % 64 == 0
