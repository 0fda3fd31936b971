"""Glow composition — mirrors reference normalizing_flow/glow.py: StepFlow :12-63, GlowBlock :66-137, Glow :140-246.

MI355X execution plan for Glow.transform (forward + log-dets):
  * the NCHW input is converted once to NHWC; each level owns ONE NHWC buffer and every flow updates it in
    place (the reference allocates ~25 temporaries per flow);
  * per level, log|det W| (and W^-1 for invert) of all K flows comes from ONE batched fp64 launch;
  * per flow: [ActNorm -> 1x1 conv] fused in one HBM pass, three MFMA convs with fused epilogues, one coupling
    launch that also adds the flow's batch-independent log-det terms — 5 launches instead of ~25 torch ops and
    no host synchronisation (the reference's `.item()` per ActNorm per call is replaced by a host-side flag).
"""
import os
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn
from torch import Tensor

import nfdp_hip as K
from nfdp_hip import graphs as _graphs
from .base import Transform, acc_in, acc_out
from .transforms import ActNorm, InvConv2d, AffineCoupling, Squeeze, Split
from .utils import get_item


class StepFlow(Transform):
    """actnorm -> invconv -> affine coupling  (reference glow.py:12-63)."""

    def __init__(self, in_channels: int = 3, coupling_net_n_features: int = 512, compute: Optional[str] = None):
        super().__init__()
        self.actnorm = ActNorm(in_channels=in_channels)
        self.invconv2d = InvConv2d(in_channels=in_channels)
        self.affcoupling = AffineCoupling(in_channels=in_channels, n_features=coupling_net_n_features, compute=compute)

    def transform_nhwc_(self, xh: Tensor, ldj: Optional[Tensor], logabsdet: Optional[Tensor] = None,
                        defer_logdet: bool = False, head_done=None, next_head=None, ldj_defer=None):
        """In place on xh.  logabsdet: this flow's [1] fp32 entry of the level's batched log|det| launch;
        defer_logdet=True leaves the 1x1-conv log-det out (the caller adds the level's sum at the end).
        head_done / next_head (the level loop, _run_flows_fwd): head_done = (const,) says the previous flow's launch already applied this
        flow's head (const: its [2] log-det constants or None); next_head = the next flow's head parameters, handed to the coupling
        (AffineCoupling.transform_nhwc_).  With next_head given the call returns (xh, head_done for the next flow | None)."""
        an, ic = self.actnorm, self.invconv2d
        if head_done is not None:
            out = self.affcoupling.transform_nhwc_(xh, ldj, head_done[0], next_head=next_head, ldj_defer=ldj_defer)
            return out if next_head is None else ((out[0], (out[1],)) if isinstance(out, tuple) else (out, None))
        if not an.initialized():
            an.init_from(xh)
        packed = self.affcoupling._fused_pack(xh)
        if packed is not None and not K.coupling_fused_has_head(xh.shape[1], xh.shape[2], xh.shape[3]):
            packed = None                         # coupling-only geometry (C = 192): head as its own launch, then the fused coupling
        if packed is not None and (ldj is None or defer_logdet or logabsdet is not None):
            # whole flow step in one launch; a per-flow log|det| (if not deferred) is pre-scaled to H*W*log|det W|
            term = None
            if ldj is not None and logabsdet is not None:
                term = K.flow_ldj_consts(an.scale.view(-1), logabsdet, xh.shape[1] * xh.shape[2])[1:]
            K.stepflow_fused(xh, packed, an.scale.view(-1), an.bias.view(-1), ic._w(), ldj, term, inverse=False)
            return xh if next_head is None else (xh, None)
        const = None
        if ldj is not None:
            if logabsdet is None and not defer_logdet:
                logabsdet, _ = K.invconv_prepare(ic._w(), want_inverse=False)
            _, const = K.flow_head_fwd(xh, an.scale.view(-1), an.bias.view(-1), ic._w(), logabsdet, out=xh)
        else:
            K.actnorm_invconv_fwd(xh, an.scale.view(-1), an.bias.view(-1), ic._w(), out=xh)
        out = self.affcoupling.transform_nhwc_(xh, ldj, const, next_head=next_head, ldj_defer=ldj_defer)
        return out if next_head is None else ((out[0], (out[1],)) if isinstance(out, tuple) else (out, None))

    def head_params(self, ldj: Optional[Tensor], logabsdet: Optional[Tensor], defer_logdet: bool):
        """(scale, bias, W, logabsdet | None) for a neighbour's fused pass, or None while the ActNorm awaits its data-dependent init
        or the log-det term would have to be computed here (no level table)."""
        an, ic = self.actnorm, self.invconv2d
        if not an.initialized() or (ldj is not None and logabsdet is None and not defer_logdet):
            return None
        return an.scale.view(-1), an.bias.view(-1), ic._w(), (logabsdet if ldj is not None else None)

    def invert_nhwc_(self, yh: Tensor, winv: Optional[Tensor] = None) -> Tensor:
        an, ic = self.actnorm, self.invconv2d
        if winv is None:
            _, winv = K.invconv_prepare(ic._w(), want_inverse=True)
        packed = self.affcoupling._fused_pack(yh)
        if packed is not None and K.coupling_fused_has_head(yh.shape[1], yh.shape[2], yh.shape[3]):
            K.stepflow_fused(yh, packed, an.scale.view(-1), an.bias.view(-1), winv.reshape(-1), None, None, inverse=True)
            return yh
        out = self.affcoupling.invert_nhwc_(yh, head=(an.scale.view(-1), an.bias.view(-1), winv.reshape(-1)))
        if isinstance(out, tuple):                # the tap + coupling pass applied the inverse head too
            return yh
        return K.invconv_actnorm_inv(yh, an.scale.view(-1), an.bias.view(-1), winv, out=yh)

    def transform(self, x: Tensor, log_det_jac: Tensor, logp: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        w = acc_in(log_det_jac)
        y = self.transform_nhwc_(K.to_nhwc(x).contiguous().clone(), w)
        return K.as_nchw(y), acc_out(log_det_jac, w), logp

    def invert(self, y: Tensor) -> Tensor:
        return K.as_nchw(self.invert_nhwc_(K.to_nhwc(y).contiguous().clone()))


# log|det W| and W^-1 depend only on the weights.  The reference recomputes them on every call (transforms.py:131, :144:
# SURVEY Appendix A #4); with this switch on they are kept per level until a 1x1-conv weight changes (tensor version /
# storage / optimizer epoch) - evaluation and sampling run many batches on fixed weights, and for a 5-level model the 16
# Gauss-Jordans of the C = 192 level cost 12 ms, a third of a forward pass.  bench.py turns it off: every timed step
# recomputes them, as the reference does.  Never read or filled while a stream is capturing.  Like the packed-weight caches the
# key is (storage, tensor version, optimizer epoch): an in-place edit through `.data` (which has its own version counter) is
# invisible to it - call nfdp_hip.bump_param_epoch() after such an edit, or switch the cache off.
CACHE_WEIGHT_TERMS = True
# Glow.transform_nhwc: the Split priors' log-density terms on a side stream, joined at the end.  Measured (round 4, L3 K16 B = 512): the
# step got 8 % SLOWER (1.90 -> 2.07 ms) - a level launch wants every CU at once, and the prior convs' workgroups in its way delay it by
# more than they take on their own.  Off.
PRIORS_ON_SIDE_STREAM = False
LOGDETS_ON_SIDE_STREAM = True      # Glow.transform_nhwc of 4- and 5-level models: the 1x1-conv log-dets on a side stream (see there)
LOGDETS_SIDE_MIN_LEVELS = 4        # ... models with at least this many levels
# Glow.transform_nhwc: Split (prior conv + log-density + the latent's copy) and the next level's Squeeze as ONE launch per level boundary
# (csrc/boundary.hip) where the image fits a workgroup's LDS, and the levels' weight-only log-det terms added by one launch: 22 -> 13
# launches per forward + NLL step of a 3-level model.  False: the per-op launches (tests compare the two).
FUSED_LEVEL_BOUNDARY = True
# ... where it pays: the boundary launch is one workgroup per IMAGE, so a small batch of large images (B = 32 at 32x32x24: 32 workgroups of
# 1024 pixels each, 70 us) is slower than the four per-op launches it replaces (measured, CelebA shape: 5.163 vs 5.140 ms per forward).
# Taken when the batch fills the chip's CUs halfway or an image is at most 64 pixels.
FUSED_BOUNDARY_MIN_BATCH = 128


def set_weight_term_cache(on: bool) -> None:
    global CACHE_WEIGHT_TERMS
    CACHE_WEIGHT_TERMS = bool(on)


def _cached_terms(flows: Sequence[StepFlow], what: str, compute):
    use = CACHE_WEIGHT_TERMS and not torch.is_grad_enabled() and not (
        torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
    if not use:
        return compute()
    key = (what, K.param_epoch()) + tuple((f.invconv2d._w().data_ptr(), f.invconv2d._w()._version) for f in flows)
    slot = flows[0].__dict__.setdefault("_weight_terms", {})
    hit = slot.get(what)
    if hit is None or hit[0] != key:
        hit = (key, compute())
        slot[what] = hit
    return hit[1]


def _level_logdets(flows: Sequence[StepFlow]) -> Tensor:
    """log|det W| of all K flows of a level in ONE fp64 launch (the reference runs K slogdets, transforms.py:131)."""
    def compute():
        w = torch.stack([f.invconv2d._w() for f in flows])                      # [K,C,C] gather (plumbing)
        return K.invconv_prepare(w, want_inverse=False)[0]
    return _cached_terms(flows, "logdets", compute)


INVERSES_IN_ONE_LAUNCH = os.environ.get("NFDP_INVERSES_IN_ONE_LAUNCH", "1") != "0"


def _model_inverses(groups) -> list:
    """W^-1 of several levels (lists of flows): the levels whose size the one-wave Gauss-Jordan takes in ONE launch
    (K.invconv_inverses), the others level by level (_level_inverses); -> one [K, C, C] tensor per level, in order."""
    small = [i for i, g in enumerate(groups) if INVERSES_IN_ONE_LAUNCH and K.invconv_inverses_supported(g[0].invconv2d._w().shape[-1])]
    out = [None] * len(groups)
    if len(small) > 1:
        flat = [f for i in small for f in groups[i]]
        got = _cached_terms(flat, "inverses_multi", lambda: K.invconv_inverses([[f.invconv2d._w() for f in groups[i]] for i in small]))
        for i, w in zip(small, got):
            out[i] = w
    for i, g in enumerate(groups):
        if out[i] is None:
            out[i] = _level_inverses(g)
    return out


def _level_inverses(flows: Sequence[StepFlow]) -> Tensor:
    """W^-1 of all K flows of a level (fp64 Gauss-Jordan, one launch; the reference inverts per call, transforms.py:144)."""
    def compute():
        w = torch.stack([f.invconv2d._w() for f in flows])
        return K.invconv_prepare(w, want_inverse=True)[1]
    return _cached_terms(flows, "inverses", compute)


# Flows per persistent launch: by default a whole level (the kernel takes up to 32 flows per launch and the ABI entry
# chunks longer levels).  A launch then holds every CU for ~1 ms, so nothing on another stream can interleave with it;
# Glow.transform_nhwc therefore computes all 1x1-conv log-dets in one short launch on the same stream beforehand.
LEVEL_CHUNK = int(os.environ.get("NFDP_LEVEL_CHUNK", "32"))


def _level_tables(flows: Sequence[StepFlow], xh: Tensor):
    """(packed images, ActNorm scales, biases) of every flow of a level when ALL of them can run on the fused kernel
    (bf16 mode, supported geometry, ActNorms initialised), else None."""
    packed = []
    if not K.level_fused_takes_heads(xh.shape[1], xh.shape[2], xh.shape[3]):
        return None                               # (C = 192 on the one-workgroup kernels: the fused launch is the coupling layer alone - flow by flow)
    for f in flows:
        if not f.actnorm.initialized():
            return None
        p = f.affcoupling._fused_pack(xh)
        if p is None:
            return None
        packed.append(p)
    return packed, [f.actnorm.scale.view(-1) for f in flows], [f.actnorm.bias.view(-1) for f in flows]


def _run_flows_fwd(flows: Sequence[StepFlow], xh: Tensor, ldj: Optional[Tensor], lads: Optional[Tensor] = None,
                   defer_logdet: bool = False) -> Tensor:
    if ldj is None or defer_logdet:
        tabs = _level_tables(flows, xh)
        if tabs is not None:                      # persistent launches, the tile stays in LDS across LEVEL_CHUNK flows
            ws = [f.invconv2d._w() for f in flows]
            for i in range(0, len(flows), LEVEL_CHUNK):
                j = i + LEVEL_CHUNK
                K.level_fused(xh, tabs[0][i:j], tabs[1][i:j], tabs[2][i:j], ws[i:j], ldj, inverse=False)
            return xh
    if ldj is not None and lads is None and not defer_logdet:
        lads = _level_logdets(flows)
    done = None                                   # (const,): the previous flow's launch applied this flow's head already
    deferred = [] if ldj is not None else None    # log-det sums of the head-fused passes: added to ldj by ONE launch behind the loop
    for i, f in enumerate(flows):
        lad = None if lads is None else lads[i:i + 1]
        nxt = None
        if i + 1 < len(flows):                    # the next flow's head rides in this flow's tap + coupling pass where that pass exists
            nxt = flows[i + 1].head_params(ldj, None if lads is None else lads[i + 1:i + 2], defer_logdet)
        if nxt is None:
            if deferred:                          # before this flow adds its own term: ldj sees the flows' terms in flow order
                K.ldj_finish_level(ldj, deferred)
            xh = f.transform_nhwc_(xh, ldj, lad, defer_logdet, head_done=done)
            done = None
        else:
            xh, done = f.transform_nhwc_(xh, ldj, lad, defer_logdet, head_done=done, next_head=nxt, ldj_defer=deferred)
    if deferred:
        K.ldj_finish_level(ldj, deferred)
    return xh


_SIDE_STREAMS = {}


def _side_stream(device, main=None) -> "torch.cuda.Stream":
    """The auxiliary HIP stream for weight-only work that overlaps the flow kernels of a pass issued on `main` (default: the
    current stream).  One per (device, ISSUING stream): two passes issued on two streams (two threads, two eager pipelines) do not
    serialise on one shared side stream, and a side stream never belongs to two captures."""
    dev = torch.device(device)
    if main is None:
        main = torch.cuda.current_stream(dev)
    key = (dev.index, main.cuda_stream)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


class _SideBranch:
    """`with _SideBranch(device, on) as br:` - br.side is the side stream of the current stream, forked behind everything the
    current stream holds so far (None when `on` is false).  `br.join(tensors)` makes the issuing stream wait for it and hands it
    the tensors the side stream produced; leaving the block ALWAYS joins (also on an exception between fork and join: an un-joined
    fork under torch.cuda.graph would surface as a capture error that hides the real one)."""

    def __init__(self, device, on: bool):
        # (under capture a side stream may only be forked from the capture's origin stream: a nested fork kills hipStreamEndCapture,
        # nfdp_hip.graphs.side_branch_allowed; elsewhere the side work is issued inline on the calling stream)
        on = bool(on) and _graphs.side_branch_allowed(device)
        self.main = torch.cuda.current_stream(device) if on else None
        self.side = _side_stream(device, self.main) if on else None
        self.joined = not on

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.main)
        return self

    def join(self, tensors=()):
        if not self.joined:
            self.main.wait_stream(self.side)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.main)
            self.joined = True

    def __exit__(self, *exc):
        if not self.joined:
            self.main.wait_stream(self.side)
            self.joined = True
        return False


def _run_flows_inv(flows: Sequence[StepFlow], yh: Tensor, winvs: Optional[Tensor] = None) -> Tensor:
    if winvs is None:
        winvs = _level_inverses(flows)
    tabs = _level_tables(flows, yh)
    if tabs is not None:
        wl = [winvs[i] for i in range(len(flows))]
        for j in range(len(flows), 0, -LEVEL_CHUNK):
            i = max(j - LEVEL_CHUNK, 0)
            K.level_fused(yh, tabs[0][i:j], tabs[1][i:j], tabs[2][i:j], wl[i:j], None, inverse=True)
        return yh
    for i in reversed(range(len(flows))):
        yh = flows[i].invert_nhwc_(yh, winvs[i])
    return yh


class GlowBlock(Transform):
    """squeeze -> K flows -> split  (reference glow.py:66-137)."""

    def __init__(self, in_channels: int = 3, K: int = 32, learn_prior_mean_logs: bool = True,
                 compute: Optional[str] = None):
        super().__init__()
        flow_channels = 4 * in_channels
        self.squeeze = Squeeze()
        self.flows = nn.ModuleList([StepFlow(in_channels=flow_channels, compute=compute) for _ in range(K)])
        self.split = Split(flow_channels, learn_prior_mean_logs=learn_prior_mean_logs, compute=compute)

    def transform_nhwc(self, xh: Tensor, ldj: Optional[Tensor], logp: Optional[Tensor], lads: Optional[Tensor] = None,
                       defer_logdet: bool = False, prior_stream=None, squeezed_in: bool = False, squeeze_out: bool = False):
        """-> (kept half, z).  squeezed_in: xh is already this level's squeezed buffer (the previous block's fused boundary wrote it);
        squeeze_out: where the geometry allows, Split + the NEXT block's Squeeze run as one launch (K.split_prior_squeeze) and the
        call returns (next level's squeezed buffer, z, True) - else (kept-half view, z, False)."""
        yh = xh if squeezed_in else K.squeeze(xh)    # fresh level buffer; flows then run in place
        yh = _run_flows_fwd(self.flows, yh, ldj, lads, defer_logdet)
        if squeeze_out:
            if FUSED_LEVEL_BOUNDARY and prior_stream is None and K.split_prior_squeeze_supported(yh) and (
                    yh.shape[0] >= FUSED_BOUNDARY_MIN_BATCH or yh.shape[1] * yh.shape[2] <= 64):
                z, ynext = self.split.transform_squeeze_nhwc(yh, logp)
                return ynext, z, True
            yv, z = self.split.transform_nhwc(yh, logp, side=prior_stream)
            return yv, z, False
        yv, z = self.split.transform_nhwc(yh, logp, side=prior_stream)
        return yv, z

    def invert_nhwc(self, yh: Tensor, zh: Optional[Tensor], temperature: float = 1.0, eps: Optional[Tensor] = None,
                    winvs: Optional[Tensor] = None) -> Tensor:
        xh = self.split.invert_nhwc(yh, zh, temperature, eps)
        xh = _run_flows_inv(self.flows, xh, winvs)
        return K.unsqueeze(xh)

    def transform(self, x: Tensor, log_det_jac: Tensor, logp: Tensor):
        wl, wp = acc_in(log_det_jac), acc_in(logp)
        yv, z = self.transform_nhwc(K.to_nhwc(x), wl, wp)
        return K.as_nchw(yv), acc_out(log_det_jac, wl), K.as_nchw(z), acc_out(logp, wp)

    def invert(self, y: Tensor, latent: Tensor = None, temperature: float = 1.0, eps: Optional[Tensor] = None) -> Tensor:
        zh = None if latent is None else K.to_nhwc(latent)
        eh = None if eps is None else K.to_nhwc(eps).contiguous()
        return K.as_nchw(self.invert_nhwc(K.to_nhwc(y), zh, temperature, eh))


class Glow(Transform):
    """The Glow model  (reference glow.py:140-246).  `compute`: 'f32' | 'bf16' | None (library default)."""

    def __init__(self, in_channel: int = 3, L: int = 3, K: int = 32, learn_prior_mean_logs: bool = True,
                 compute: Optional[str] = None):
        super().__init__()
        self.L = L
        self.K = K
        self.in_channel = in_channel
        self.blocks = nn.ModuleList(GlowBlock(in_channels=(2 ** i * self.in_channel), K=self.K,
                                              learn_prior_mean_logs=learn_prior_mean_logs, compute=compute)
                                    for i in range(self.L - 1))
        self.final_squeeze = Squeeze()
        self.final_flows = nn.ModuleList(
            StepFlow(in_channels=(2 ** (self.L + 1) * self.in_channel), compute=compute) for _ in range(self.K)
        )

    def set_compute(self, compute: Optional[str]) -> None:
        for m in self.modules():
            if hasattr(m, "compute"):
                m.compute = compute

    def _check_no_grad(self):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError(
                "nfdp HIP path: only Glow.transform has a backward (normalizing_flow/training.py); call invert/sample "
                "under torch.no_grad() or freeze the parameters, results would otherwise silently carry no gradient.")

    def transform_nhwc(self, xh: Tensor, ldj: Optional[Tensor], logp: Optional[Tensor]):
        """-> ([z_0 .. z_{L-1}] NHWC tensors, ldj, logp); accumulators fp64, updated in place."""
        zs = []
        yh = xh
        # The 1x1-conv log-dets depend only on the weights: ONE launch computes all K*L of them (fp64 LU, one workgroup
        # per matrix, ~10-20 us) ahead of the flow kernels; the per-level sums  H*W * sum_k log|det W_k|  are folded into
        # ldj once at the end (an fp64 re-association of the reference's per-flow adds, transforms.py:131).
        lads = None
        # Deep models (L >= 4): the wide levels' Gauss-Jordans are not short (1.4 ms for the 16 C = 192 matrices of a 5-level model: a
        # fifth of its forward pass) and nothing reads their result before the end, so they run on a side stream beside the first
        # levels' flows - parallel branches of a captured graph.  (L = 3: 25 us, and a level launch holds every CU: same stream.)
        with _SideBranch(xh.device, ldj is not None and LOGDETS_ON_SIDE_STREAM and self.L >= LOGDETS_SIDE_MIN_LEVELS and xh.is_cuda) as br:
            if ldj is not None:
                with torch.cuda.stream(br.side if br.side is not None else torch.cuda.current_stream(xh.device)):
                    groups = [list(b.flows) for b in self.blocks] + [list(self.final_flows)]
                    small = [g for g in groups if g[0].invconv2d._w().shape[-1] <= 96]      # the one-launch LU holds a matrix in LDS
                    flat = _cached_terms([f for g in small for f in g], "logdets_all",
                                         lambda: K.invconv_logdets([f.invconv2d._w() for g in small for f in g])) if small else None
                    lads, o = [], 0
                    for g in groups:
                        if g[0].invconv2d._w().shape[-1] <= 96:
                            lads.append(flat[o:o + len(g)])
                            o += len(g)
                        else:                      # C = 192 at the last level of a 5-level model: batched Gauss-Jordan, global workspace
                            lads.append(_level_logdets(g))
            hws = []
            # the Split priors' terms (a narrow 3x3 conv + the Gaussian log-density per level: ~25 us each, launch-bound) only feed
            # logp: on a second stream they run beside the next level's flows instead of between two level launches
            pstream = None
            if PRIORS_ON_SIDE_STREAM and logp is not None and xh.is_cuda and _graphs.side_branch_allowed(xh.device):
                pstream = _side_stream(xh.device) if br.side is None else br.side
            try:
                squeezed = False
                for i, block in enumerate(self.blocks):
                    yh, z, squeezed = block.transform_nhwc(yh, ldj, logp, None, defer_logdet=True, prior_stream=pstream,
                                                           squeezed_in=squeezed, squeeze_out=True)
                    hws.append(z.shape[1] * z.shape[2])
                    zs.append(z)
                if not squeezed:
                    yh = K.squeeze(yh)
                hws.append(yh.shape[1] * yh.shape[2])
                yh = _run_flows_fwd(self.final_flows, yh, ldj, None, defer_logdet=True)
                zs.append(yh)
            finally:
                if pstream is not None and pstream is not br.side:
                    torch.cuda.current_stream(xh.device).wait_stream(pstream)
            if ldj is not None:
                br.join(lads)
                if FUSED_LEVEL_BOUNDARY and len(lads) <= 8:
                    K.ldj_add_levels(ldj, list(zip(lads, hws)))
                else:
                    for lad, hw in zip(lads, hws):
                        K.ldj_add_scaled(ldj, lad, lad.numel(), hw)
        return zs

    def transform(self, x: Tensor, log_det_jac: Tensor, logp: Tensor) -> Tuple[list, Tensor, Tensor]:
        """(reference glow.py:172-201) -> (latent parts, log_det_jac, logp); logp may be None (NFBackbone path).
        When gradients are required the differentiable exact-fp32 path runs (training.py: autograd Functions over the
        HIP forward/backward kernels, nothing in place); otherwise the fused in-place inference path."""
        from . import training
        if training.grad_needed(self):
            return training.glow_transform(self, x, log_det_jac, logp)
        wl, wp = acc_in(log_det_jac), acc_in(logp)
        zs = self.transform_nhwc(K.to_nhwc(x), wl, wp)
        return [K.as_nchw(z) for z in zs], acc_out(log_det_jac, wl), acc_out(logp, wp)

    def invert_nhwc(self, latents: list, temperature: float = 1.0, eps: Optional[Sequence[Tensor]] = None) -> Tensor:
        xh = latents[-1].contiguous().clone()                 # flows run in place; keep the caller's latent intact
        # Deep models (L >= 4): the deepest level runs first and needs its W^-1 at once (the sixteen C = 192 Gauss-Jordans: 1.2 ms), but
        # the other levels' inversions (0.5 ms together for a 5-level model) are not needed until their level starts: a side stream
        # computes them beside the deepest level's work - a parallel branch of a captured graph, as the log-dets of the forward pass.
        pre, fin = None, None
        with _SideBranch(xh.device, LOGDETS_ON_SIDE_STREAM and self.L >= 4 and xh.is_cuda) as br:
            deep = br.side is not None
            if deep:
                with torch.cuda.stream(br.side):
                    pre = _model_inverses([list(b.flows) for b in self.blocks])
            elif xh.is_cuda:
                # shallow models: the 1x1 convs of ALL levels inverted by one launch in front of the pass (one wave per matrix; a
                # 3-level model ran three launches, one in front of each level: 94 + 20 + 20 us -> as long as the largest)
                allinv = _model_inverses([list(b.flows) for b in self.blocks] + [list(self.final_flows)])
                pre, fin = allinv[:-1], allinv[-1]
            xh = _run_flows_inv(self.final_flows, xh, fin)
            xh = K.unsqueeze(xh)
            if deep:
                br.join(pre)
        j = 0
        for i, block in enumerate(reversed(self.blocks)):
            latent = get_item(latents, -(i + 2))              # None -> Split samples from its learned prior
            e = None
            if latent is None and eps is not None:
                e = eps[j]
                j += 1
            xh = block.invert_nhwc(xh, latent, temperature, e, winvs=None if pre is None else pre[len(self.blocks) - 1 - i])
        return xh

    def invert(self, latents: list, temperature: float = 1.0, eps: Optional[Sequence[Tensor]] = None) -> Tensor:
        """(reference glow.py:203-228).  eps: optional list of N(0,1) tensors (logical NCHW) for the Splits that
        sample, in visiting order (deepest block first)."""
        self._check_no_grad()
        lat = [K.to_nhwc(z) for z in latents]
        eh = None if eps is None else [K.to_nhwc(e).contiguous() for e in eps]
        return K.as_nchw(self.invert_nhwc(lat, temperature, eh))

    # BASELINE.json's north_star spells the call surface `Glow.forward() / inverse()`; the reference defines neither (its Glow has only
    # transform / invert, glow.py:172, :203; SURVEY 8(b) allows the aliases).  `flow(x, ldj, logp)` therefore works like transform.
    def forward(self, x: Tensor, log_det_jac: Tensor, logp: Tensor) -> Tuple[list, Tensor, Tensor]:
        return self.transform(x, log_det_jac, logp)

    def inverse(self, latents: list, temperature: float = 1.0, eps: Optional[Sequence[Tensor]] = None) -> Tensor:
        return self.invert(latents, temperature, eps)

    @torch.no_grad()
    def sample(self, latents: list, postprocess_func=None, temperature: float = 1.0) -> Tensor:
        """(reference glow.py:230-246)"""
        self.eval()
        new_samples = self.invert(latents, temperature)
        final_output = postprocess_func(new_samples.float()) if postprocess_func else new_samples.float()
        self.train()
        return final_output
