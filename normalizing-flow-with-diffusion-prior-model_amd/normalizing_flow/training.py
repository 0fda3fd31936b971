"""Differentiable (training) path of the flow: torch.autograd.Function wrappers over the HIP kernels.

The reference trains through torch autograd (normalizing_flow/trainer.py:152-167: transform -> calculate_loss ->
loss.backward() -> clips -> optimizer.step()).  Here every arithmetic op of that graph is a Function whose forward
and backward are HIP launches through the C ABI (the module's compute mode - exact-fp32 or bf16 MFMA - applies to the
convolutions' forward, data-gradient and weight-gradient GEMMs; activations stay fp32; include/nfdp.h "backward of the flow
path"); autograd only strings them together.  What stays in torch is glue of parameter size or smaller: the
`H*W*sum(scale)` / `H*W*log|det W|` scalars added to the per-sample log-det, the [2C] prior vector
`bias*exp(3*logs)`, channel slices and the per-sample adds of the fp64 accumulators.

Layout: NHWC fp32 activations like the inference path, parameters in the reference's layouts; nothing is updated in
place (autograd needs the inputs), so this path is selected only when gradients are required.
"""
import os
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

import nfdp_hip as K


def grad_needed(module: torch.nn.Module) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


# Parameter gradients straight into .grad: where a parameter already holds a dense fp32 .grad the launch that finishes a parameter
# gradient ADDS it there and the node returns None for it, instead of handing autograd a temporary that AccumulateGrad adds with one
# more elementwise launch per parameter (~640 per step of Glow L3 K16, plus the flip / permute copies that brought the GEMM results
# into OIHW order).  Because the node returns None, the parameter's AccumulateGrad node never runs - and neither does anything hanging
# on it.  Hooks registered from Python (tensor hooks, post-accumulate-grad hooks: nfdp_hip.dist.BucketedGradAllReduce) are visible and
# switch the parameter back to the ordinary path, but torch's DistributedDataParallel registers its reducer hooks in C++ on the
# AccumulateGrad node, where they cannot be seen.  The path is therefore OPT-IN per parameter: only the owner of the gradient buffer
# marks a parameter (`enable_direct_grads`; nfdp_hip.optim.FusedAdam(direct_grads=...) does, and by default only when no process group
# with more than one rank exists), stating that no external reducer depends on AccumulateGrad.  DIRECT_PARAM_GRADS = False switches it
# off globally (torch.autograd.grad() on parameters that hold a .grad needs that).
DIRECT_PARAM_GRADS = True


def enable_direct_grads(params, on: bool = True) -> None:
    """Mark parameters whose gradients the backward launches may add straight into their dense .grad (see above)."""
    for p in params:
        p._nfdp_direct_grad = bool(on)


def _slot(p):
    if not DIRECT_PARAM_GRADS or not isinstance(p, torch.nn.Parameter) or not p.requires_grad:
        return None
    if not getattr(p, "_nfdp_direct_grad", False):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or g.shape != p.shape or not g.is_contiguous() or not g.is_cuda:
        return None
    if p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
        return None
    return g


# Weight-gradient GEMMs on a side stream.  The backward of a flow is a chain of data-gradient launches, each waiting for the one before
# (coupling -> zero conv -> hidden layer 2 -> hidden layer 1 -> input conv -> 1x1 conv -> ActNorm -> next flow); at the 8x8 / 4x4
# levels most of them sit at their ~5-14 us launch floor with the chip nearly idle.  The weight-gradient GEMMs and their split-K
# reduces only produce parameter gradients - nothing later in the chain reads them - so they are forked onto a second HIP stream
# (behind everything enqueued so far) and joined once per level / node; under a captured step they become parallel branches of the
# hipGraph.  Tensors they read are kept referenced until the join, so the caching allocator cannot hand their memory to a
# main-stream launch while the side stream still reads it.
WGRAD_SIDE_STREAM = __import__("os").environ.get("NFDP_WGRAD_SIDE_STREAM", "1") != "0"
# Activation recomputation for the fused training path (reference trainer.py:163-167 keeps every activation): 0 = keep the
# hidden activations of all flows (fastest, 9.7 GiB at B = 512 L3 K16); G > 0 = keep only the flow-state tensors and, in the backward
# pass, re-run the training forward of G flows at a time from the group's saved input (bit-identical activations, so bit-identical
# gradients; one extra forward per step; the hidden tensors of G flows live at a time).  A flow is invertible, so its input COULD also be
# recomputed from its output - not done: the flow state is 6 MB per flow here, the hidden activations are 268 MB.
# A dict {H: G} sets G per level by its spatial size (levels not named keep their activations): {16: 8} recomputes only the 16x16 level,
# which holds 4.5 of the 6.9 GiB, in two groups.
RECOMPUTE_FLOWS = 0


def _recompute_group(h: int) -> int:
    r = RECOMPUTE_FLOWS
    if isinstance(r, dict):
        return int(r.get(int(h), 0))
    return int(r) if r else 0


# The main stream waits for the side stream's weight gradients every SIDE_JOIN_EVERY flows of a level's backward pass (0: once, at the
# end of the level).  Tensors the side stream reads - a flow's hidden-activation gradients, 0.3 GB at level 0 - are kept referenced
# until the join; joining every few flows bounds that to a few flows' worth instead of a whole level's 4.5 GB.
import os as _os
SIDE_JOIN_EVERY = int(_os.environ.get("NFDP_SIDE_JOIN_EVERY", "2"))      # measured at B = 512 L3 K16: 13.9 ms / 9.7 GiB at 0, 13.6 ms / 6.9 GiB at 2
_side_streams = {}
_side_keep = []


_side_jobs = []


def _defer_side(fn, *tensors) -> None:
    """Queue a weight-gradient launch for the side stream (run by _flush_side at the end of the flow: ONE fork per flow - every fork
    and join is a cross-queue dependency in the captured graph); without the side stream it runs at once."""
    if not WGRAD_SIDE_STREAM:
        fn()
    else:
        _side_jobs.append((fn, tensors))


def _flush_side() -> None:
    if _side_jobs:
        jobs = list(_side_jobs)
        _side_jobs.clear()
        with _OnSide(*[t for _, ts in jobs for t in ts]):
            for fn, _ in jobs:
                fn()


class _OnSide:
    """`with _OnSide(t0, t1, ...):` - launches inside go to the side stream, ordered after the main stream's work so far."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        if not WGRAD_SIDE_STREAM:
            return self
        from nfdp_hip.graphs import side_branch_allowed
        if not side_branch_allowed():         # capturing on a stream that is not the capture's origin: a nested fork kills
            return self                       # hipStreamEndCapture (nfdp_hip.graphs) - the weight gradients run inline instead
        dev = torch.cuda.current_device()
        side = _side_streams.get(dev)
        if side is None:
            side = _side_streams[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        _side_keep.append(self.tensors)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _join_side() -> None:
    """The main stream waits for the side stream's weight gradients (before anything may read them: autograd's accumulation of
    returned gradients, the optimizer, a gradient all-reduce)."""
    if _side_keep:
        side = _side_streams.get(torch.cuda.current_device())
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        _side_keep.clear()


_join_queued = [False]


def _join_at_end_of_backward() -> None:
    """Queue ONE engine callback per backward pass that makes the main stream wait for the side stream (and releases the tensors it
    reads): runs after the last node, before loss.backward() returns to the caller."""
    if _join_queued[0]:
        return
    _join_queued[0] = True

    def done():
        _join_queued[0] = False
        _join_side()

    torch.autograd.Variable._execution_engine.queue_callback(done)


def _abandon_side() -> None:
    """A backward launch raised between _defer_side and _flush_side / _join_side: drop the queued closures (they would otherwise run
    in the NEXT backward pass and add stale weight gradients into .grad) and let the main stream wait for whatever the side stream
    was already given before the tensors it reads are released."""
    _side_jobs.clear()
    _join_queued[0] = False
    try:
        _join_side()
    except Exception:                                        # noqa: BLE001 - the original error is the one to report
        _side_keep.clear()


def _side_guarded(backward):
    """Decorator for the backward of a node that uses the side stream: on any exception the queue is abandoned."""
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        try:
            return backward(ctx, *grads)
        except BaseException:
            _abandon_side()
            raise
    return wrapped


def _slots(*params):
    """The .grad tensors of all params, or None unless every one of them can be accumulated into directly."""
    gs = [_slot(p) for p in params]
    return None if any(g is None for g in gs) else gs


# ----------------------------------------------------------------------------- Functions
class ActNormFn(Function):
    """y = exp(scale) * (x + bias)   (reference transforms.py:80)."""

    @staticmethod
    def forward(ctx, xh, scale, bias):
        y = K.actnorm_fwd(xh.contiguous(), scale.detach().reshape(-1), bias.detach().reshape(-1))
        ctx.save_for_backward(y, scale)
        ctx.params = (scale, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        gs = _slots(*ctx.params)
        if gs is not None:
            dx, _, _ = K.scale_bwd(dy.contiguous(), y, scale.detach().reshape(-1), 1.0, False, grads_out=(gs[0].view(-1), gs[1].view(-1)))
            return dx, None, None
        dx, dscale, dbias = K.scale_bwd(dy.contiguous(), y, scale.detach().reshape(-1), 1.0, False)
        return dx, dscale.view_as(scale), dbias.view_as(scale)


class InvConvFn(Function):
    """y = conv2d(x, W) with W [C,C,1,1]   (reference transforms.py:132).  wt: W^T [C,C] for the data gradient, from the level's
    one batched transpose (else transposed here)."""

    @staticmethod
    def forward(ctx, xh, weight, wt=None):
        c = weight.shape[0]
        xh = xh.contiguous()
        y = K.invconv_apply(xh, weight.detach().reshape(c, c).contiguous())
        ctx.save_for_backward(xh, weight, wt)
        ctx.params = (weight,)
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, weight, wt = ctx.saved_tensors
        c = weight.shape[0]
        dy = dy.contiguous()
        if wt is None:
            wt = weight.detach().reshape(c, c).t().contiguous()
        dx = K.invconv_apply(dy, wt)                                # W^T dy
        gs = _slots(*ctx.params)
        if gs is not None and weight.dim() == 4:
            K.conv_wgrad(xh, dy, 1, out=gs[0], accumulate=True)
            return dx, None, None
        dw = K.conv_wgrad(xh, dy, 1)
        return dx, dw.view_as(weight), None


class LogAbsDetFn(Function):
    """log|det W| in fp64 on the device, returned as fp32 [1]   (reference transforms.py:131); gradient g * W^-T.
    `pre` = (lad [1], winv [C,C]) from a batched launch over all flows of a level (else computed here)."""

    @staticmethod
    def forward(ctx, weight, pre):
        c = weight.shape[0]
        if pre is None:
            lad, winv = K.invconv_prepare(weight.detach().reshape(1, c, c).contiguous(), want_inverse=True)
            winv = winv[0]
        else:
            lad, winv = pre
        ctx.save_for_backward(winv)
        ctx.wshape = weight.shape
        return lad.clone()

    @staticmethod
    def backward(ctx, g):
        (winv,) = ctx.saved_tensors
        return (g.reshape(()) * winv.t()).reshape(ctx.wshape), None


class ConvScaleFn(Function):
    """y = act((conv(x, W) + add) * exp(f * mul)): Conv2dActNorm (+ReLU) with add/mul = ActNorm bias/scale, f = 1
    (reference utils.py:67-69, :83-87) and ZeroConv2d with add/mul = bias/logs, f = 3 (utils.py:43-44).  One fused
    MFMA launch forward; backward = one scale/ReLU-backward pass (+ per-channel parameter sums), the data gradient as
    the same conv kernel on flipped/transposed weights, and the weight-gradient GEMM over pixels.  `compute` (K.F32 exact
    fp32 MFMA | K.BF16 bf16 MFMA with fp32 accumulation) applies to all three GEMMs; activations stay fp32."""

    @staticmethod
    def forward(ctx, xh, weight, add, mul, f, relu, packed, k, compute):
        # bf16 mode: the 512-channel hidden activations (the ReLU outputs) are stored in bf16 - they are bf16 MFMA operands
        # of the next conv and of its weight gradient anyway - which halves the HBM traffic that dominates the step
        hidden_bf16 = compute == K.BF16 and bool(relu)
        xcol = None
        if K.use_narrow_conv(weight.shape[1], weight.shape[0], k):       # 512 -> C: pointwise GEMM + tap shift-sum
            y = K.conv3x3_narrow(xh, weight, add=add.detach().reshape(-1), mul=mul.detach().reshape(-1), mul_exp=float(f),
                                 relu=bool(relu), compute=compute)
        elif K.use_im2col_conv(weight.shape[1], weight.shape[0], k, compute):   # C/2 -> 512: explicit im2col + pointwise GEMM
            xcol = K.im2col3x3(xh)                                        # (the packed 3x3 weight is the 1x1 weight for its rows)
            y = K.conv2d(xcol, packed, weight.shape[0], 1, add=add.detach().reshape(-1), mul=mul.detach().reshape(-1),
                         mul_exp=float(f), relu=bool(relu), compute=compute, out_dtype=K.BF16 if hidden_bf16 else K.F32)
        else:
            y = K.conv2d(xh, packed, weight.shape[0], k, add=add.detach().reshape(-1), mul=mul.detach().reshape(-1),
                         mul_exp=float(f), relu=bool(relu), compute=compute, out_dtype=K.BF16 if hidden_bf16 else K.F32)
        ctx.save_for_backward(xh if xcol is None else xcol, weight, mul, y)
        ctx.im2col = xcol is not None
        ctx.x_dtype = xh.dtype
        ctx.f, ctx.relu, ctx.k, ctx.compute = float(f), bool(relu), int(k), compute
        ctx.add_shape = add.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        xh, weight, mul, y = ctx.saved_tensors
        dc, dmul, dadd = K.scale_bwd(dy.contiguous(), y, mul.detach().reshape(-1), ctx.f, ctx.relu)   # dc in y's storage type
        dx, dc_col = None, None
        if ctx.needs_input_grad[0]:
            wd = weight.detach().contiguous()           # the data-gradient weight (flipped taps, channels swapped) is packed straight
            if K.use_narrow_conv(weight.shape[0], weight.shape[1], ctx.k):   # from the parameter.  C/2 -> 512: its data gradient is a 512 -> C/2 conv
                dx = K.conv3x3_narrow(dc, None, compute=ctx.compute, co=weight.shape[1],
                                      packed_taps=K.conv_pack_ex(wd, ctx.compute, flip=True, taps=True))
            elif dc.dtype == torch.float32 and K.use_im2col_conv(weight.shape[0], weight.shape[1], ctx.k, ctx.compute):
                # data gradient of the 512 -> C zero conv: a C -> 512 conv of the narrow dc, as im2col + pointwise GEMM
                dc_col = K.im2col3x3(dc)
                dx = K.conv2d(dc_col, K.conv_pack_ex(wd, ctx.compute, flip=True), weight.shape[1], 1, compute=ctx.compute,
                              out_dtype=K.BF16 if ctx.x_dtype == torch.bfloat16 else K.F32)
            else:
                dx = K.conv2d(dc, K.conv_pack_ex(wd, ctx.compute, flip=True), weight.shape[1], ctx.k, compute=ctx.compute,
                              out_dtype=K.BF16 if ctx.x_dtype == torch.bfloat16 else K.F32)
            if dx.dtype != ctx.x_dtype:
                dx = dx.to(ctx.x_dtype)
        if dc_col is not None and xh.dtype == torch.bfloat16:
            # weight gradient of the zero conv from the same im2col matrix: G[i][tap*C + c] = sum_p x[p][i] dc[p + shift(tap)][c] is a
            # plain GEMM over pixels of two bf16 tensors (LDS-DMA kernel, no gather), and dW[c][i][ky][kx] = G[i][2-ky][2-kx][c]
            o, i = weight.shape[:2]
            g = K.conv_wgrad(dc_col, xh, 1, compute=ctx.compute)[:, :9 * o, 0, 0]
            dw = g.reshape(i, 3, 3, o).flip(1, 2).permute(3, 0, 1, 2).contiguous()
        elif ctx.im2col:                                                   # xh is the saved im2col matrix: a plain GEMM over pixels
            o, i = weight.shape[:2]
            dw = K.conv_wgrad(xh, dc, 1, compute=ctx.compute)[:, :9 * i, 0, 0].reshape(o, 3, 3, i).permute(0, 3, 1, 2).contiguous()
        else:
            dw = K.conv_wgrad(xh, dc, ctx.k, compute=ctx.compute)
        return dx, dw, dadd.reshape(ctx.add_shape), dmul.view_as(mul), None, None, None, None, None


def _coupling_forward(ctx, xh, hh, ldj, own_ldj):
    """Shared by CouplingFn / CouplingStepFn: y = coupling(x, h) out of place, log-det into the fp64 accumulator (ldj itself when
    the caller owns it - a fresh fp64 tensor of this graph, marked dirty - else a copy)."""
    y = xh.clone()
    if ldj is None:
        acc = torch.zeros(xh.shape[0], device=xh.device, dtype=torch.float64)
    elif own_ldj and ldj.dtype == torch.float64 and ldj.is_contiguous():
        acc = ldj
        ctx.mark_dirty(ldj)
    else:
        acc = ldj.to(torch.float64, copy=True)
    K.coupling_fwd(y, hh, acc, None)
    ctx.ldj_dtype = None if ldj is None else ldj.dtype
    return y, acc


def _ldj_grad(ctx, dacc):
    if ctx.ldj_dtype is None or dacc is None:
        return None
    return dacc if dacc.dtype == ctx.ldj_dtype else dacc.to(ctx.ldj_dtype)


class CouplingFn(Function):
    """(x_a | x_b), (log_s | t), ldj -> (x_a | (x_b + t) * sigmoid(log_s + 2)), ldj + per-sample sum log(s + 1e-6)
    (reference transforms.py:179-185).  The running log-det goes through the node (the kernel adds into it), so the chain of flows
    costs no separate add launch; ldj=None returns the per-sample increment alone."""

    @staticmethod
    def forward(ctx, xh, hh, ldj, own_ldj=False):
        xh, hh = xh.contiguous(), hh.contiguous()
        y, acc = _coupling_forward(ctx, xh, hh, ldj, own_ldj)
        ctx.save_for_backward(xh, hh)
        return y, acc

    @staticmethod
    def backward(ctx, dy, dacc):
        xh, hh = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(xh)
        dx, dh = K.coupling_bwd(xh, hh, dy.contiguous(), None if dacc is None else dacc.contiguous())
        return dx, dh, _ldj_grad(ctx, dacc), None


class LevelLogdetFn(Function):
    """hw * sum over the flows of a level of (sum(actnorm.scale) + log|det W|) as ONE node (reference transforms.py:82, :131-133,
    added per flow there): one batched fp64 Gauss-Jordan for the log-dets and inverses, one product for all dW = g hw W^-T.
    Per flow the same terms cost ~14 scalar-sized launches forward + backward (sum, two products, three adds, their reductions)."""

    @staticmethod
    def forward(ctx, hw, n, *params):
        ws, scales = params[:n], params[n:]
        c = ws[0].shape[0]
        w = torch.stack([x.detach().reshape(c, c) for x in ws])
        lads, winvs = K.invconv_prepare(w, want_inverse=True)
        ssum = torch.cat([x.detach().reshape(-1) for x in scales]).sum(dtype=torch.float64)
        ctx.save_for_backward(winvs)
        ctx.hw, ctx.n = float(hw), n
        ctx.wshape, ctx.sshape = ws[0].shape, scales[0].shape
        ctx.params = params
        return ((lads.sum(dtype=torch.float64) + ssum) * float(hw)).reshape(1)

    @staticmethod
    def backward(ctx, g):
        (winvs,) = ctx.saved_tensors
        n = ctx.n
        gh = (g.reshape(()) * ctx.hw).to(torch.float32)
        dw = (gh * winvs.transpose(1, 2)).contiguous()
        dws = [dw[i].view(ctx.wshape) for i in range(n)]
        gs = _slots(*ctx.params)
        if gs is not None:                                         # two multi-tensor launches instead of 2n accumulations
            torch._foreach_add_(gs[:n], dws)
            torch._foreach_add_(gs[n:], [gh.expand(ctx.sshape).contiguous()] * n)
            return (None,) * (2 + 2 * n)
        return (None, None) + tuple(dws) + (gh.expand(ctx.sshape),) * n


class GaussLogpFn(Function):
    """Per-sample log N(z; mean, exp(logs)) with h = (mean | logs) per pixel (h_mode 0) or per channel (h_mode 1)
    (reference prior.py:36-37)."""

    @staticmethod
    def forward(ctx, zh, h, h_mode):
        inc = torch.zeros(zh.shape[0], device=zh.device, dtype=torch.float64)
        h = h.contiguous()
        K.gauss_logp(zh, h, h_mode, inc)
        ctx.save_for_backward(zh, h)
        ctx.h_mode = h_mode
        return inc

    @staticmethod
    def backward(ctx, dinc):
        zh, h = ctx.saved_tensors
        dz, dh = K.gauss_logp_bwd(zh, h, ctx.h_mode, dinc.contiguous())
        if ctx.h_mode == 1:
            _, _, dh = K.scale_bwd(dh, want_dc=False)            # sum over batch and pixels -> [2C]
            dh = dh.view_as(h)
        return dz, dh, None


class SqueezeFn(Function):
    """Space-to-depth by 2 (reference transforms.py:226); the gradient is the inverse reshuffle."""

    @staticmethod
    def forward(ctx, xh):
        return K.squeeze(xh)

    @staticmethod
    def backward(ctx, dy):
        return K.unsqueeze(dy.contiguous())


# ----------------------------------------------------------------------------- module walkers
def _conv_actnorm_relu(mod, xh: Tensor, c: int) -> Tensor:
    """Conv2dActNorm followed by the ReLU of coupling_network (utils.py:83-87)."""
    an, cv = mod.actnorm, mod.conv
    packed = cv.packed(c)
    if not an.initialized():
        with torch.no_grad():                                     # data-dependent init on the raw conv output
            an.init_from(K.conv2d(xh.detach(), packed, cv.out_channels, mod.filter_size, compute=c, out_dtype=K.F32))
    return ConvScaleFn.apply(xh, cv.weight, an.bias, an.scale, 1.0, True, packed, mod.filter_size, c)


def _zero_conv(mod, xh: Tensor, c: int) -> Tensor:
    return ConvScaleFn.apply(xh, mod.weight, mod.bias, mod.logs, mod.logscale_factor, False,
                             mod._cache.get(mod.weight, c), mod.filter_size, c)


def _net_forward(xa, w1, b1, s1, w2, b2, s2, w3, b3, l3, f3, packs):
    """The coupling network (utils.py:72-90) in bf16 training mode: im2col + pointwise GEMM, pointwise GEMM, tap GEMM + shift-sum; the
    hidden activations stay in bf16.  -> h and what the backward needs."""
    c = K.BF16
    p1, p2, p3t = packs                                        # packed w1 (3x3 = im2col form), w2 (1x1), tap form of w3
    xcol = K.im2col3x3(xa)
    y1 = K.conv2d(xcol, p1, w1.shape[0], 1, add=b1.detach().reshape(-1), mul=s1.detach().reshape(-1), mul_exp=1.0, relu=True,
                  compute=c, out_dtype=K.BF16)
    y2 = K.conv2d(y1, p2, w2.shape[0], 1, add=b2.detach().reshape(-1), mul=s2.detach().reshape(-1), mul_exp=1.0, relu=True,
                  compute=c, out_dtype=K.BF16)
    h = K.conv3x3_narrow(y2, None, add=b3.detach().reshape(-1), mul=l3.detach().reshape(-1), mul_exp=float(f3), relu=False,
                         compute=c, packed_taps=p3t, co=w3.shape[0])
    return h, (xcol, y1, y2)


def _net_backward(dh, h, xcol, y1, y2, params, f3, need_dxa, bpacks=None):
    """-> (dxa or None, the nine parameter gradients in `params` order or None when they went straight into .grad).  The scale/ReLU
    backward of the two hidden layers runs in the EPILOGUE of the GEMM that produces their output gradient
    (nfdp_pointwise_dgrad_masked_bf16): dy1 and dy2 never go to memory and y1, y2 are read once there - separately that pass moved
    0.4 GB per hidden layer and flow at level 0 (96 launches, 3.8 ms per step)."""
    w1, b1, s1, w2, b2, s2, w3, b3, l3 = params
    c = K.BF16
    co, hid = w3.shape[0], w3.shape[1]
    o1, i1 = w1.shape[:2]
    if bpacks is None:                                         # data-gradient layouts: from the step's batched packing, else packed here
        bpacks = (K.conv_pack_ex(w3.detach().contiguous(), c, flip=True), K.conv_pack_ex(w2.detach().contiguous(), c, flip=True),
                  K.conv_pack_ex(w1.detach().contiguous(), c, flip=True, taps=True) if need_dxa else None)
    gs = _slots(*params)                                       # parameter gradients added straight into .grad (see DIRECT_PARAM_GRADS)
    direct = gs is not None
    if direct:
        gw1, gb1, gs1, gw2, gb2, gs2, gw3, gb3, gl3 = gs
    flat = lambda a, b: (a.view(-1), b.view(-1))
    # zero conv: h = (u3 + b3) e^{3 l3}
    dc3, dl3, db3 = K.scale_bwd(dh.contiguous(), h, l3.detach().reshape(-1), f3, False, grads_out=flat(gl3, gb3) if direct else None)
    dc3_col = K.im2col3x3(dc3)
    if direct:                                                  # dW3 from the same im2col matrix, stored mirrored into OIHW by the reduce
        _defer_side(lambda: K.conv_wgrad(dc3_col, y2, 1, compute=c, out=gw3, accumulate=True, form=2, form_c=co), dc3_col, y2)
    else:
        g = K.conv_wgrad(dc3_col, y2, 1, compute=c)[:, :9 * co, 0, 0]
        dw3 = g.reshape(hid, 3, 3, co).flip(1, 2).permute(3, 0, 1, 2).contiguous()
    # hidden layer 2: its output gradient is the C -> 512 conv of dc3; its scale/ReLU backward rides in that GEMM's epilogue
    dc2, ds2, db2 = K.pointwise_dgrad_masked(dc3_col, bpacks[0], y2,
                                             s2.detach().reshape(-1), 1.0, grads_out=flat(gs2, gb2) if direct else None)
    if direct:
        _defer_side(lambda: K.conv_wgrad(y1, dc2, 1, compute=c, out=gw2, accumulate=True), y1, dc2)
        dw2 = None
    else:
        dw2 = K.conv_wgrad(y1, dc2, 1, compute=c)
    # hidden layer 1: output gradient = dc2 W2 (1x1), same fusion
    dc1, ds1, db1 = K.pointwise_dgrad_masked(dc2, bpacks[1], y1,
                                             s1.detach().reshape(-1), 1.0, grads_out=flat(gs1, gb1) if direct else None)
    if direct:
        _defer_side(lambda: K.conv_wgrad(xcol, dc1, 1, compute=c, out=gw1, accumulate=True, form=1, form_c=i1), xcol, dc1)
    else:
        dw1 = K.conv_wgrad(xcol, dc1, 1, compute=c)[:, :9 * i1, 0, 0].reshape(o1, 3, 3, i1).permute(0, 3, 1, 2).contiguous()
    dxa = None
    if need_dxa:                                                # 512 -> C/2 data gradient: tap GEMM + shift-sum
        dxa = K.conv3x3_narrow(dc1, None, compute=c, co=i1, packed_taps=bpacks[2])
    if direct:
        return dxa, None
    return dxa, (dw1, db1.view_as(b1), ds1.view_as(s1), dw2, db2.view_as(b2), ds2.view_as(s2), dw3, db3.view_as(b3), dl3.view_as(l3))


class CouplingStepFn(Function):
    """The affine coupling of a flow step as ONE autograd node for the bf16 training mode (reference transforms.py:170-186 with the
    network of utils.py:72-90): h = NN(x_a), (x_a | (x_b + t) s), ldj + sum log s.  One node instead of network + slice + coupling
    nodes: the data gradient of the network is added into the first half of the coupling's dx in place, where autograd's slice backward
    costs a zero fill, a copy and an add of full-size tensors per flow."""

    @staticmethod
    def forward(ctx, xh, ldj, own_ldj, f3, packs, bpacks, *params):
        xh = xh.contiguous()
        c = xh.shape[-1]
        h, (xcol, y1, y2) = _net_forward(xh[..., : c // 2], *params, f3, packs)
        y, acc = _coupling_forward(ctx, xh, h, ldj, own_ldj)
        ctx.save_for_backward(xh, h, xcol, y1, y2)
        ctx.params, ctx.f3, ctx.bpacks = params, float(f3), bpacks
        return y, acc

    @staticmethod
    @_side_guarded
    def backward(ctx, dy, dacc):
        xh, h, xcol, y1, y2 = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(xh)
        dx, dh = K.coupling_bwd(xh, h, dy.contiguous(), None if dacc is None else dacc.contiguous())
        dxa, grads = _net_backward(dh, h, xcol, y1, y2, ctx.params, ctx.f3, ctx.needs_input_grad[0], ctx.bpacks)
        if dxa is not None:
            dx[..., : dxa.shape[-1]] += dxa
        _flush_side()
        _join_side()
        return (dx, _ldj_grad(ctx, dacc), None, None, None, None) + (grads if grads is not None else (None,) * 9)


def _fused_net_shapes_ok(net) -> bool:
    c1, c2, zc = net[0], net[2], net[4]
    hid, c = c1.conv.out_channels, K.BF16
    return (c1.filter_size == 3 and c2.filter_size == 1 and zc.filter_size == 3
            and K.use_im2col_conv(c1.conv.in_channels, hid, 3, c) and K.use_im2col_conv(zc.out_channels, hid, 3, c)
            and K.use_narrow_conv(hid, zc.out_channels, 3) and K.use_narrow_conv(hid, c1.conv.in_channels, 3))


def _fused_net_ok(net, xa: Tensor, c: int) -> bool:
    if c != K.BF16 or xa.dtype != torch.float32:
        return False
    c1, c2, zc = net[0], net[2], net[4]
    npix = xa.shape[0] * xa.shape[1] * xa.shape[2]
    hid = c1.conv.out_channels
    return (c1.actnorm.initialized() and c2.actnorm.initialized() and _fused_net_shapes_ok(net)
            and K.pointwise_dgrad_masked_supported(npix, K.im2col_kpad(zc.out_channels), hid)
            and K.pointwise_dgrad_masked_supported(npix, hid, hid))


class _StepPacks:
    """The six kernel layouts per coupling network the bf16 training step needs (w1 im2col form, w2, w3 tap form; data-gradient
    forms of w3, w2 and the tap form of w1's), for all flows of a model, refreshed by one launch per step (K.PackBatch) instead of six
    launches per flow."""

    def __init__(self, flows):
        specs, self.index = [], {}
        for f in flows:
            net = f.affcoupling.net
            if K.compute_enum(f.affcoupling.compute) != K.BF16 or not _fused_net_shapes_ok(net):
                continue
            w1, w2, w3 = net[0].conv.weight, net[2].conv.weight, net[4].weight
            self.index[id(f)] = len(specs)
            specs += [(w1, False, False), (w2, False, False), (w3, False, True), (w3, True, False), (w2, True, False), (w1, True, True)]
        self.batch = K.PackBatch(specs, K.BF16) if specs else None        # the Parameters themselves: matches() follows re-pointed .data

    def refresh(self) -> None:
        """A new step: the layouts are stale.  The packing launch itself waits for the first get() - a model whose levels all run on the
        fused level kernels (their own images) never asks, and 288 layouts cost 137 us at the head of every step."""
        self._fresh = False

    def get(self, flow):
        i = self.index.get(id(flow))
        if i is None:
            return None
        if not getattr(self, "_fresh", False):
            self.batch.run()
            self._fresh = True
        return self.batch.views[i:i + 6]


def _step_packs(glow):
    flows = [f for b in glow.blocks for f in b.flows] + list(glow.final_flows)
    sp = getattr(glow, "_step_packs", None)
    key = tuple(id(f) for f in flows) + tuple(f.affcoupling.compute for f in flows)
    if sp is None or sp.key != key or (sp.batch is not None and not sp.batch.matches()):
        sp = _StepPacks(flows)
        sp.key = key
        object.__setattr__(glow, "_step_packs", sp)
    sp.refresh()
    return sp


def coupling_net(net, xa: Tensor, compute: Optional[str] = None) -> Tensor:
    """The coupling network as three conv nodes (any mode, any size)."""
    c = K.compute_enum(compute)
    return _zero_conv(net[4], _conv_actnorm_relu(net[2], _conv_actnorm_relu(net[0], xa, c), c), c)


def step_flow(flow, xh: Tensor, ldj: Optional[Tensor], pre=None, wt: Optional[Tensor] = None, own_ldj: bool = False, six=None):
    """actnorm -> invconv -> affine coupling (reference glow.py:44-49), differentiable.  pre="level": the weight-only log-det terms of
    this flow (actnorm scale sum, log|det W|) are added once per level by the caller (LevelLogdetFn); wt: W^T of the 1x1 conv;
    own_ldj: ldj is a tensor of this graph the coupling node may add into in place; six: the flow's layouts from _StepPacks."""
    an, ic, cp = flow.actnorm, flow.invconv2d, flow.affcoupling
    hw = xh.shape[1] * xh.shape[2]
    if not an.initialized():
        with torch.no_grad():
            an.init_from(xh.detach().contiguous())
    y = ActNormFn.apply(xh, an.scale, an.bias)
    y = InvConvFn.apply(y, ic.weight, wt)
    c = y.shape[-1]
    level = isinstance(pre, str)
    net = cp.net
    if _fused_net_ok(net, y[..., : c // 2], K.compute_enum(cp.compute)):
        c1, c2, zc = net[0], net[2], net[4]
        cc = K.BF16
        if six is not None:                                        # this step's batched packing (forward + data-gradient layouts)
            packs, bpacks = tuple(six[:3]), tuple(six[3:])
        else:
            packs, bpacks = (c1.conv.packed(cc), c2.conv.packed(cc), zc._cache.get_taps(zc.weight, cc)), None
        y, inc = CouplingStepFn.apply(y, ldj if level else None, own_ldj and level, zc.logscale_factor, packs, bpacks, c1.conv.weight,
                                      c1.actnorm.bias, c1.actnorm.scale, c2.conv.weight, c2.actnorm.bias, c2.actnorm.scale, zc.weight,
                                      zc.bias, zc.logs)
    else:
        h = coupling_net(net, y[..., : c // 2], cp.compute)
        y, inc = CouplingFn.apply(y, h, ldj if level else None, own_ldj and level)
    if ldj is None:
        return y, None
    if level:
        return y, inc
    return y, ldj + hw * an.scale.sum() + hw * LogAbsDetFn.apply(ic.weight, pre) + inc


# ----------------------------------------------------------------------------- a whole level as one node (fused forward)
FUSED_TRAIN_FORWARD = True          # bf16 mode, 32x32-model geometries: the level's forward is ONE persistent launch (see LevelFusedFn)


FUSED_TRAIN_BACKWARD = True         # ... and the data-gradient chain of each coupling network one launch (nfdp_coupling_bwd_fused_bf16)
# the three weight gradients of a coupling network as one GEMM launch + one reduce launch (nfdp_coupling_wgrad3_bf16); False: one
# conv_wgrad_affine / conv_wgrad_blocked call per gradient (A/B measurements, tests)
WGRAD_ONE_LAUNCH = os.environ.get("NFDP_WGRAD_ONE_LAUNCH", "1") != "0"


# The weight gradients of the 8x8 / 4x4 levels' flows as ONE launch per level (K.coupling_wgrad3_batch) instead of one per flow: a single
# flow's three GEMMs cover 32768 / 8192 pixels of a B = 512 step - 66 / 52 us per flow in isolation against 165 us for the 16x larger
# 16x16 flow, 1.9 ms of a 13.1 ms step.  The flows' operands (hidden-activation gradients: 1.2 / 0.3 GB per level) are kept until the
# level's data-gradient chain is through; the launch then runs on the side stream beside the NEXT level's chain and is joined by that
# level (or, for the last node of the pass, by an engine callback before backward() returns).  Levels with more pixels per flow than
# this keep the per-flow launches (their operands are 0.3 GB per flow, and one flow fills the chip).
WGRAD_BATCH_MAX_PIXELS = int(os.environ.get("NFDP_WGRAD_BATCH_MAX_PIXELS", str(64 * 512)))
# Memory: the kept operands are 82 MB per 8x8 flow of a B = 512 step (21 MB at 4x4).  A level's flows are therefore sent in groups of
# WGRAD_BATCH_FLOWS (the previous group joined first, which releases its operands: 0.65 GB at a time at the 8x8 level); a level of at most
# WGRAD_BATCH_OVERLAP_PIXELS pixels per flow leaves its last group running beside the next node's chain (smaller values join it at the end
# of its node, so nothing of it is alive when the next level allocates).  Measured, B = 512 L3 K16 (profiles/r05_m_wgrad_batch_sweep.txt):
# groups of 8 + overlap 12.56 ms at 7.67 GiB peak; groups of 8 joined at the node's end 13.01 ms at 7.67; groups of 4 13.16 ms at 7.22;
# groups of 2 15.5 ms at 7.07; per-flow launches (round 4) 13.36 ms at 6.84 GiB.
WGRAD_BATCH_FLOWS = int(os.environ.get("NFDP_WGRAD_BATCH_FLOWS", "8"))
WGRAD_BATCH_OVERLAP_PIXELS = int(os.environ.get("NFDP_WGRAD_BATCH_OVERLAP_PIXELS", str(64 * 512)))


def _net_backward_fused(dc3, dx, xcol, y1, y2, params, net, packed=None, batch=None, masks=None):
    """Backward of one coupling network on the fused kernels, from the gradient dc3 at the zero conv's raw output (the link kernel has
    applied its scale backward).  dx [B,H,W,C]: the coupling's input gradient, whose first half receives += the network's input
    gradient; xcol: saved im2col rows of the network's input; y1, y2: saved hidden activations (blocked layout, [Bpad,H,W,512] bf16).
    One launch for the data path (coupling_bwd_fused: zero conv^T -> mask/scale -> 1x1^T -> mask/scale -> input conv^T, hidden
    gradients written once, im2col rows of dc3 on the side), three GEMMs over pixels for the parameters - the two hidden layers'
    ActNorm gradients come out of their weight-gradient GEMMs (conv_wgrad_affine).  -> gradients of (w1, b1, s1, w2, b2, s2, w3) in
    that order, or None when they went straight into .grad."""
    w1, b1, s1, w2, b2, s2, w3, b3, l3 = params
    co, hid = w3.shape[0], w3.shape[1]
    o1, i1 = w1.shape[:2]
    gs = _slots(w1, b1, s1, w2, b2, s2, w3)
    direct = gs is not None
    if packed is None:                                     # (LevelFusedFn packs the whole level's images in one launch)
        packed = K.coupling_bwd_fused_pack(w1.detach(), s1.detach().reshape(-1), w2.detach(), s2.detach().reshape(-1), w3.detach(),
                                           out=getattr(net, "_fused_bwd", None) if net is not None else None)
        if net is not None:
            object.__setattr__(net, "_fused_bwd", packed)
    dc1, dc2, dc3_col = K.coupling_bwd_fused(dc3, dx, packed, y1, y2, masks=masks)     # (masks: the ReLU signs of y1 / y2 at 1/8 of their bytes)
    def problems(gw1, gb1, gs1, gw2, gb2, gs2, gw3):
        # the three weight gradients as ONE GEMM launch + one reduce (coupling_wgrad3): the 512 x 512 one first, it takes longest
        return [dict(x=y1, dy=dc2, dw=gw2, x_blocked=True, weight=w2.detach(), bias=b2.detach().reshape(-1), dscale=gs2.view(-1),
                     dbias=gb2.view(-1)),
                dict(x=xcol, dy=dc1, dw=gw1, form=1, form_c=i1, weight=w1.detach().contiguous(), bias=b1.detach().reshape(-1),
                     dscale=gs1.view(-1), dbias=gb1.view(-1)),
                dict(x=dc3_col, dy=y2, dw=gw3, form=2, form_c=co)]
    if direct:
        pr = problems(*gs)
        if WGRAD_ONE_LAUNCH and batch is not None and WGRAD_SIDE_STREAM:
            batch.append(pr)                                # (the dicts keep the operands alive until the level's one launch)
        elif WGRAD_ONE_LAUNCH and batch is not None:
            K.coupling_wgrad3_batch([pr], True)
        elif WGRAD_ONE_LAUNCH:
            _defer_side(lambda: K.coupling_wgrad3(pr, True), dc3_col, y2, y1, dc2, xcol, dc1)
        else:
            gw1, gb1, gs1, gw2, gb2, gs2, gw3 = gs
            _defer_side(lambda: K.conv_wgrad_blocked(dc3_col, y2, gw3, True, form=2, form_c=co, dy_blocked=True), dc3_col, y2)
            _defer_side(lambda: K.conv_wgrad_affine(y1, dc2, w2.detach(), b2.detach().reshape(-1), gw2, gs2.view(-1), gb2.view(-1), True,
                                                    x_blocked=True, dy_blocked=True), y1, dc2)
            _defer_side(lambda: K.conv_wgrad_affine(xcol, dc1, w1.detach(), b1.detach().reshape(-1), gw1, gs1.view(-1), gb1.view(-1), True,
                                                    form=1, form_c=i1, dy_blocked=True), xcol, dc1)
        return None
    dw3 = torch.empty_like(w3)
    dw2, ds2, db2 = torch.empty_like(w2), torch.empty(hid, device=w2.device), torch.empty(hid, device=w2.device)
    dw1, ds1, db1 = torch.empty_like(w1), torch.empty(o1, device=w1.device), torch.empty(o1, device=w1.device)
    if WGRAD_ONE_LAUNCH and batch is not None:
        # the small levels' slice plan (the same bits as the level-wide launch gives this flow: the plan does not depend on the batch)
        K.coupling_wgrad3_batch([problems(dw1, db1, ds1, dw2, db2, ds2, dw3)], False)
    elif WGRAD_ONE_LAUNCH:
        K.coupling_wgrad3(problems(dw1, db1, ds1, dw2, db2, ds2, dw3), False)
    else:
        K.conv_wgrad_blocked(dc3_col, y2, dw3, False, form=2, form_c=co, dy_blocked=True)
        K.conv_wgrad_affine(y1, dc2, w2.detach(), b2.detach().reshape(-1), dw2, ds2, db2, False, x_blocked=True, dy_blocked=True)
        K.conv_wgrad_affine(xcol, dc1, w1.detach().contiguous(), b1.detach().reshape(-1), dw1, ds1, db1, False, form=1, form_c=i1,
                            dy_blocked=True)
    return (dw1, db1.view_as(b1), ds1.view_as(s1), dw2, db2.view_as(b2), ds2.view_as(s2), dw3)


class LevelFusedFn(Function):
    """All K flow steps of a Glow level (reference glow.py:107-109: actnorm -> invconv -> affine coupling, K times) as ONE autograd
    node whose forward is ONE persistent launch of the fused flow-step kernel (nfdp_level_fused_train_bf16): the flow state stays in
    LDS across the flows and the 512-wide hidden activations are produced in registers; the launch writes out, per flow, exactly what
    the backward launches read (flow input, coupling input, zero-conv output, the two hidden activations in bf16).  The unfused
    training forward cost 8.1 ms of a 26.5 ms step (L3 K16, B=512) in ~1000 launches; this is 3 launches.  Backward: per flow, in
    reverse, the same launches the per-flow nodes issued (coupling_bwd, _net_backward, 1x1-conv and ActNorm backward).

    Inputs: xh [B,H,W,C] (not modified), ldj fp64 [B] or None, meta = (packed images, W^T stack, bpacks per flow, f3), then 12
    parameters per flow: actnorm.scale, actnorm.bias, invconv.weight, w1, b1, s1, w2, b2, s2, w3, b3, l3.  The weight-only log-det terms
    are NOT included (LevelLogdetFn)."""

    NP = 12

    @staticmethod
    def forward(ctx, xh, ldj, meta, *params):
        packed, wts, bpacks, f3 = meta[:4]
        n = len(packed)
        c = xh.shape[-1]
        y = xh.contiguous().clone()
        acc = torch.zeros(xh.shape[0], device=xh.device, dtype=torch.float64) if ldj is None else ldj.to(torch.float64, copy=True)
        P = LevelFusedFn.NP
        scales = [params[P * i].detach().reshape(-1) for i in range(n)]
        biases = [params[P * i + 1].detach().reshape(-1) for i in range(n)]
        ws = [params[P * i + 2].detach().reshape(c, c) for i in range(n)]
        ctx.recompute = _recompute_group(xh.shape[1])
        if ctx.recompute:
            # activation recomputation: the 512-channel hidden activations (2 x 134 MB per level-0 flow at B = 512) are not kept; the
            # backward pass re-runs this launch per group of `recompute` flows from the group's saved input state xpre[first flow]
            xpre, ypost, hout, _, _, _ = K.level_fused_train(y, packed, scales, biases, ws, acc, keep_hidden=False)
            ctx.save_for_backward(xpre, ypost, hout)
            ctx.fwd_args = (packed, scales, biases, ws)
        else:
            xpre, ypost, hout, y1, y2, xcol, masks = K.level_fused_train(y, packed, scales, biases, ws, acc, want_masks=True)
            ctx.save_for_backward(xpre, ypost, hout, y1, y2, xcol, masks)
        ctx.params, ctx.bpacks, ctx.f3, ctx.n = params, bpacks, float(f3), n
        ctx.nets = meta[4] if len(meta) > 4 else [None] * n
        ctx.ldj_dtype = None if ldj is None else ldj.dtype
        return y, acc

    @staticmethod
    @_side_guarded
    def backward(ctx, dy, dacc):
        from nfdp_hip import dist as _dist
        _dist.grads_ready(ctx.params, include=False)      # data parallelism: the later levels' gradient slices can travel now
        if ctx.recompute:
            xpre, ypost, hout = ctx.saved_tensors
            G = ctx.recompute
            packed_f, scales_f, biases_f, ws_f = ctx.fwd_args
            cache = {}

            def hidden(i):
                """(xcol, y1, y2) of flow i: the training forward of its group of G flows, re-run from the group's saved input."""
                g0 = (i // G) * G
                if cache.get("g0") != g0:
                    if cache:
                        flush_batch()                              # (small levels: the previous group's weight gradients in one launch)
                        _flush_side()                              # the side stream still reads the previous group's tensors (weight
                        _join_side()                               # gradients): join before they are released
                    cache.clear()                                  # drop the previous group's tensors before allocating the next
                    g1 = min(g0 + G, ctx.n)
                    _, _, _, y1g, y2g, xcolg, mg = K.level_fused_train(xpre[g0].clone(), packed_f[g0:g1], scales_f[g0:g1], biases_f[g0:g1],
                                                                       ws_f[g0:g1], None, want_masks=True)
                    cache.update(g0=g0, t=(xcolg, y1g, y2g, mg))
                xc, a1, a2, mg = cache["t"]
                return xc[i - g0], a1[i - g0], a2[i - g0], (None if mg is None else mg[i - g0])
        else:
            xpre, ypost, hout, y1, y2, xcol, masks = ctx.saved_tensors

            def hidden(i):
                return xcol[i], y1[i], y2[i], (None if masks is None else masks[i])
        n, P, params = ctx.n, LevelFusedFn.NP, ctx.params
        bsz, hh, ww, c = ypost.shape[1:]
        npix = bsz * hh * ww
        # small levels: the flows' weight gradients in one launch behind the chain (under recomputation: per group of flows)
        batch = [] if (WGRAD_ONE_LAUNCH and npix <= WGRAD_BATCH_MAX_PIXELS) else None

        def flush_batch():
            if batch:
                _join_side()                                   # the previous group (or level) is through: its operands are released
                prs = list(batch)
                batch.clear()
                _defer_side(lambda: K.coupling_wgrad3_batch(prs, True), *[t for fl in prs for pr in fl for t in (pr["x"], pr["dy"])])
                _flush_side()
        if dy is None:
            dy = torch.zeros_like(ypost[0])
        dy = dy.contiguous()
        gl = None if dacc is None else dacc.contiguous()
        f3 = ctx.f3
        grads = [None] * (P * n)
        # The chain per flow, last flow first:  [link: head of flow i+1 backwards + coupling of flow i backwards] -> reverse coupling
        # network of flow i (one launch) -> ...; the weight gradients of flow i go to the side stream.  Call k of the link kernel
        # (k = 0: coupling of the last flow only; k = n: head of the first flow only) leaves per-workgroup partial sums of the four
        # small parameter gradients it touches; ONE finish launch per level reduces them all.
        part = K.flow_link_part(n + 1, npix, c, ypost.device)
        l3 = lambda i: params[P * i + 11].detach().reshape(-1)
        dx, dc3, _ = K.flow_link_bwd(dy, part[0], coupling=(ypost[n - 1], hout[n - 1], gl, l3(n - 1), f3))
        direct_small = []
        # the reverse images of the level's networks in one launch (weights are fixed during the backward pass)
        bw = [(params[P * i + 3].detach(), params[P * i + 5].detach().reshape(-1), params[P * i + 6].detach(),
               params[P * i + 8].detach().reshape(-1), params[P * i + 9].detach()) for i in range(n)]
        rev = K.coupling_bwd_fused_pack_batch(bw, outs=[getattr(nt, "_fused_bwd", None) if nt is not None else None for nt in ctx.nets])
        for nt, img in zip(ctx.nets, rev):
            if nt is not None:
                object.__setattr__(nt, "_fused_bwd", img)
        for i in reversed(range(n)):
            scale, bias, weight = params[P * i: P * i + 3]
            net = params[P * i + 3: P * i + 12]
            xc_i, y1_i, y2_i, m_i = hidden(i)
            g9 = _net_backward_fused(dc3, dx, xc_i, y1_i, y2_i, net, ctx.nets[i], packed=rev[i], batch=batch, masks=m_i)   # dx[..., :c/2] += network input gradient
            head = (xpre[i], weight.detach().reshape(c, c), scale.detach().reshape(-1), bias.detach().reshape(-1))
            cpl = (ypost[i - 1], hout[i - 1], gl, l3(i - 1), f3) if i > 0 else None
            # (the 1x1 conv's weight gradient dW = sum_p dx (x) a, transforms.py:132, is left in the call's partial buffer too)
            dx_next, dc3, _ = K.flow_link_bwd(dx, part[n - i], head=head, coupling=cpl, want_a=False)
            _flush_side()                                            # one fork per flow: the coupling network's weight gradients
            if batch is None and SIDE_JOIN_EVERY and (n - i) % SIDE_JOIN_EVERY == 0 and i > 0:
                _join_side()
            if batch is not None and len(batch) >= WGRAD_BATCH_FLOWS:
                flush_batch()
            if g9 is not None:
                grads[P * i + 3: P * i + 10] = list(g9)
            dx = dx_next
        # small parameter gradients: call k feeds ActNorm (scale, bias) of flow n-k (k >= 1) and the zero conv (logs, bias) of flow n-1-k
        small = []
        for k in range(n + 1):
            hf, cf = n - k, n - 1 - k
            small.append(((params[P * hf], params[P * hf + 1]) if k >= 1 else (None, None)) +
                         ((params[P * cf + 11], params[P * cf + 10]) if cf >= 0 else (None, None)) +
                         ((params[P * hf + 2],) if k >= 1 else (None,)))
        flat = [t for tup in small for t in tup if t is not None]
        gs = _slots(*flat)
        if gs is not None:
            it = iter(gs)
            outs = [[None if t is None else next(it).view(-1) for t in tup] for tup in small]
            acc = True
        else:
            outs = [[None if t is None else torch.empty(t.numel(), device=t.device, dtype=torch.float32) for t in tup] for tup in small]
            acc = False
        flush_batch()                                                # the level's weight gradients: one launch, on the side stream
        K.flow_link_finish(part, npix, c, f3, [o[0] for o in outs], [o[1] for o in outs], [o[2] for o in outs], [o[3] for o in outs], acc,
                           dw=[o[4] for o in outs])
        if not acc:
            for k, (tup, o) in enumerate(zip(small, outs)):
                hf, cf = n - k, n - 1 - k
                if k >= 1:
                    grads[P * hf], grads[P * hf + 1] = o[0].view_as(tup[0]), o[1].view_as(tup[1])
                    grads[P * hf + 2] = o[4].view_as(tup[4])
                if cf >= 0:
                    grads[P * cf + 11], grads[P * cf + 10] = o[2].view_as(tup[2]), o[3].view_as(tup[3])
        if batch is None or any(g is not None for g in grads) or npix > WGRAD_BATCH_OVERLAP_PIXELS:
            _join_side()
        else:
            # the level's one weight-gradient launch keeps running beside the next node's chain; whoever joins the side stream next (a
            # later level's node) - or this callback at the end of the backward pass - orders it before any consumer of the gradients
            _join_at_end_of_backward()
        if all(g is None for g in grads):                 # every gradient of this level went straight into the optimizer's buffer
            _dist.grads_ready(ctx.params, include=True, producer_streams=tuple(_side_streams.values()))
        dl = None
        if ctx.ldj_dtype is not None and dacc is not None:
            dl = dacc if dacc.dtype == ctx.ldj_dtype else dacc.to(ctx.ldj_dtype)
        return (dx, dl, None) + tuple(grads)


def _level_fused_ok(flows, yh: Tensor) -> bool:
    if not FUSED_TRAIN_FORWARD or yh.dtype != torch.float32 or len(flows) > 32:
        return False
    b, h, w, c = yh.shape
    hid = flows[0].affcoupling.net[0].conv.out_channels
    if not K.coupling_fused_supported(h, w, c, hid):
        return False
    if not (FUSED_TRAIN_BACKWARD and K.coupling_bwd_fused_supported(h, w, c, hid)):
        return False            # the saved activations are in the fused kernels' blocked layout: only the fused backward reads it
    for f in flows:
        cp = f.affcoupling
        net = cp.net
        if K.compute_enum(cp.compute) != K.BF16 or not f.actnorm.initialized():
            return False
        if not (net[0].actnorm.initialized() and net[2].actnorm.initialized()):
            return False
    return True


def _level_fused(flows, yh: Tensor, ldj: Optional[Tensor], sp, wts: Tensor):
    packed, bpacks, params = [], [], []
    nets = [f.affcoupling.net for f in flows]
    # training images of the level's networks: the stale ones (all of them after an optimizer step) re-packed in ONE launch
    keys = [n_.fused_train_key() for n_ in nets]
    stale = [i for i, n_ in enumerate(nets) if getattr(n_, "_fused_train_key", None) != keys[i]]
    if stale:
        outs = K.coupling_fused_pack_batch([nets[i].fused_pack_tensors() for i in stale], fold=False,
                                           outs=[getattr(nets[i], "_fused_train", None) for i in stale])
        for i, o in zip(stale, outs):
            nets[i]._fused_train, nets[i]._fused_train_key = o, keys[i]
    for f in flows:
        net = f.affcoupling.net
        c1, c2, zc = net[0], net[2], net[4]
        packed.append(net._fused_train)
        bpacks.append(None)                       # (the fused backward packs its own reverse images: coupling_bwd_fused_pack_batch)
        params += [f.actnorm.scale, f.actnorm.bias, f.invconv2d.weight, c1.conv.weight, c1.actnorm.bias, c1.actnorm.scale,
                   c2.conv.weight, c2.actnorm.bias, c2.actnorm.scale, zc.weight, zc.bias, zc.logs]
    f3 = flows[0].affcoupling.net[4].logscale_factor
    return LevelFusedFn.apply(yh, ldj, (packed, wts, bpacks, f3, [f.affcoupling.net for f in flows]), *params)


def _level_flows(flows, yh: Tensor, ldj: Optional[Tensor], sp=None):
    """All flows of a level; with a log-det accumulator the weight-only terms are one LevelLogdetFn node for the level, the transposed
    1x1 weights (data gradient) come from one batched transpose, and the per-sample terms are added into one fp64 tensor in place."""
    c = yh.shape[-1]
    wts = torch.stack([f.invconv2d.weight.detach().reshape(c, c) for f in flows]).transpose(1, 2).contiguous()
    hw = yh.shape[1] * yh.shape[2]
    if _level_fused_ok(flows, yh):                                # one persistent launch for the level's forward
        yh, inc = _level_fused(flows, yh, ldj, sp, wts)
        if ldj is None:
            return yh, None
        term = LevelLogdetFn.apply(hw, len(flows), *[f.invconv2d.weight for f in flows], *[f.actnorm.scale for f in flows])
        return yh, inc + term
    if ldj is None:
        for i, flow in enumerate(flows):
            yh, _ = step_flow(flow, yh, None, wt=wts[i], six=sp.get(flow) if sp else None)
        return yh, None
    ldj = ldj.to(torch.float64, copy=True)                        # this graph's own accumulator from here on
    for i, flow in enumerate(flows):
        yh, ldj = step_flow(flow, yh, ldj, "level", wts[i], True, sp.get(flow) if sp else None)  # data-dependent ActNorm init (if any) happens in here, before the scales are read
    term = LevelLogdetFn.apply(hw, len(flows), *[f.invconv2d.weight for f in flows], *[f.actnorm.scale for f in flows])
    return yh, ldj + term


def split(mod, xh: Tensor, logp: Optional[Tensor]):
    """(reference transforms.py:270-289) -> (kept half view, latent part, logp)."""
    c = xh.shape[-1]
    yv, zv = xh[..., : c // 2], xh[..., c // 2:]
    if logp is not None:
        if mod.conv is not None:
            h = _zero_conv(mod.conv, yv, K.compute_enum(mod.compute))
        else:
            h = torch.zeros(xh.shape[:3] + (c,), device=xh.device, dtype=torch.float32)
        logp = logp + GaussLogpFn.apply(zv, h, 0)
    return yv, zv.contiguous(), logp


def glow_transform(glow, x: Tensor, ldj: Optional[Tensor], logp: Optional[Tensor]):
    """Glow.transform with a gradient graph (reference glow.py:172-201): -> (latent parts NCHW views, ldj, logp)."""
    yh = K.to_nhwc(x)
    zs = []
    sp = _step_packs(glow)                                       # one launch: every weight layout this step's forward and backward read
    for block in glow.blocks:
        yh, ldj = _level_flows(block.flows, SqueezeFn.apply(yh), ldj, sp)
        yh, z, logp = split(block.split, yh, logp)
        zs.append(K.as_nchw(z))
    yh, ldj = _level_flows(glow.final_flows, SqueezeFn.apply(yh), ldj, sp)
    zs.append(K.as_nchw(yh))
    return zs, ldj, logp


def gaussian_prior_log_prob(prior, x: Tensor) -> Tensor:
    """GaussianPrior.compute_log_prob with a gradient graph (reference prior.py:70-83): the zero-input ZeroConv2d
    output is bias * exp(3 * logs) at every pixel."""
    conv = prior.conv
    c = x.shape[1]
    if conv is None:
        h = torch.zeros(2 * c, device=x.device, dtype=torch.float32)
    else:
        # the conv weight multiplies an all-zero input: its gradient is exactly zero, not absent (AdamW still decays it)
        h = conv.bias * torch.exp(conv.logs.reshape(-1) * conv.logscale_factor) + 0.0 * conv.weight.sum()
    return GaussLogpFn.apply(K.to_nhwc(x), h, 1).float()


def bits_per_dim(log_likelihood: Tensor, n_bins: float, n_pixel: float) -> Tensor:
    """calculate_loss with a gradient graph (reference utils.py:244-256); [B]-sized glue."""
    import math
    bpd_const = math.log2(math.e) / n_pixel
    return ((math.log(n_bins) * n_pixel - log_likelihood) * bpd_const).mean(dim=0)
