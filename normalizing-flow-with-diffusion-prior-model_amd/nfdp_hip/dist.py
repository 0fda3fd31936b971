"""Data-parallel host logic: one process per GPU, batch sharded across ranks, no data-path collective.

The flow treats samples independently, so forward/NLL and sampling shard over the batch with no
exchange; the only collectives are the global bits/dim — all-reduce(SUM) of the fp64 pair (sum_b bpd_b, count) —
and MAX of a timing.  Backend "nccl" is RCCL over xGMI on MI355X; "gloo" drives the CPU tests.

Replica initialisation (SURVEY.md §8e "Init"): the reference's initial state is random (`InvConv2d` draws a QR factor,
transforms.py:112-114) and data-dependent (`ActNorm`, transforms.py:74-78, through `data_dependent_nf_initialization`,
utils.py:275-292), so replicas that initialise on their own diverge from step 0.  `broadcast_module` copies rank `src`'s parameters
and buffers to every rank; `data_dependent_init` runs the reference's initialisation pass either on rank `src`'s batch alone and
broadcasts the result (mode "rank0": exactly the single-process semantics with that batch - the parity mode), or on every rank's
shard with each ActNorm's moments all-reduced as it initialises (mode "global": the statistics of the GLOBAL first batch).
"""
import contextlib
from typing import Iterable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) share of n items for `rank` (ragged tail spread over the first ranks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_bits_per_dim(pair: torch.Tensor) -> torch.Tensor:
    """pair = fp64 [2] (sum_b bpd_b, count) of this rank -> global mean bits/dim (same on all ranks)."""
    p = pair.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo":                 # CPU rehearsal backend: reduce on the host
            p = p.cpu()
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
    return p[0] / p[1]


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def _multi(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def broadcast_tensors(tensors: Iterable[torch.Tensor], src: int = 0, group=None) -> int:
    """In-place broadcast of rank `src`'s values, coalesced into ONE flat collective per dtype (a Glow L3 K16 has 726 state
    entries: 88 MB fp32 go out as one xGMI-sized message, the uint8 flags as another).  Returns the number of collectives."""
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    if not _multi(group) or not tensors:
        return 0
    host = dist.get_backend(group) == "gloo"
    n = 0
    with torch.no_grad():
        for dtype in sorted({t.dtype for t in tensors}, key=str):           # same order on every rank
            part = [t for t in tensors if t.dtype == dtype]
            flat = torch.cat([t.detach().reshape(-1) for t in part])
            dev = flat.device
            if host and flat.is_cuda:                                        # CPU rehearsal backend
                flat = flat.cpu()
            dist.broadcast(flat, src=src, group=group)
            n += 1
            flat = flat.to(dev)
            off = 0
            for t in part:
                k = t.numel()
                t.copy_(flat[off:off + k].view(t.shape))                     # bumps the tensor version: layout caches / init flags follow
                off += k
    return n


def broadcast_module(module: torch.nn.Module, src: int = 0, group=None, buffers: bool = True) -> int:
    """Every parameter (and buffer: ActNorm's `is_initialized`, the diffusion schedules) of `module` takes rank `src`'s value.
    Call it once after construction / checkpoint loading / data-dependent initialisation; a no-op without a process group."""
    ts = [p.data for p in module.parameters()]
    if buffers:
        ts += [b for b in module.buffers()]
    n = broadcast_tensors(ts, src=src, group=group)
    if n:
        from . import bump_param_epoch
        bump_param_epoch()
    return n


def merge_moments(mean: torch.Tensor, var_unbiased: torch.Tensor, count: float, group=None) -> Tuple[torch.Tensor, torch.Tensor, float]:
    """Per-channel (mean, unbiased variance, count) of this rank's rows -> those of all ranks' rows together: ONE all-reduce(SUM) of
    [n, n mean, (n-1) var + n mean^2] in fp64 (the pooled-variance identity, exact up to fp64 round-off)."""
    m, v = mean.double(), var_unbiased.double()
    n = float(count)
    pack = torch.cat([torch.tensor([n], dtype=torch.float64, device=m.device), n * m, (n - 1.0) * v + n * m * m])
    if _multi(group):
        dev = pack.device
        if dist.get_backend(group) == "gloo" and pack.is_cuda:
            pack = pack.cpu()
        dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=group)
        pack = pack.to(dev)
    c = m.numel()
    n_all = float(pack[0])
    mean_all = pack[1:1 + c] / n_all
    var_all = (pack[1 + c:] - n_all * mean_all * mean_all) / max(n_all - 1.0, 1.0)
    return mean_all, var_all.clamp_min(0.0), n_all


_ACTNORM_SYNC = {"group": None, "active": False}


def actnorm_init_sync(scale: torch.Tensor, bias: torch.Tensor, rows: int) -> None:
    """Called by ActNorm.init_from after the local init kernel when a "global" initialisation pass is running: recover this rank's
    moments from (scale, bias) = (-log(std + 1e-6), -mean) (transforms.py:76-77), merge them over the ranks and rewrite both."""
    if not _ACTNORM_SYNC["active"]:
        return
    with torch.no_grad():
        mean = -bias.double().reshape(-1)
        std = torch.exp(-scale.double().reshape(-1)) - 1e-6
        mean_all, var_all, _ = merge_moments(mean, std * std, rows, _ACTNORM_SYNC["group"])
        scale.copy_((-torch.log(var_all.sqrt() + 1e-6)).to(scale.dtype).view(scale.shape))
        bias.copy_((-mean_all).to(bias.dtype).view(bias.shape))


@contextlib.contextmanager
def global_actnorm_init(group=None):
    """Inside this context every ActNorm that initialises merges its moments over the ranks (145 small all-reduces for L3 K16,
    first batch only)."""
    _ACTNORM_SYNC.update(group=group, active=_multi(group))
    try:
        yield
    finally:
        _ACTNORM_SYNC.update(group=None, active=False)


@torch.no_grad()
def data_dependent_init(flow: torch.nn.Module, batch: torch.Tensor, src: int = 0, mode: str = "rank0", group=None,
                        n_bins: Optional[float] = None) -> None:
    """The reference's `data_dependent_nf_initialization` (utils.py:275-292: one no-grad `flow.transform` of the first batch, which
    initialises every ActNorm) for replicas.  `batch` is this rank's preprocessed first batch (dequantisation noise of
    trainer.py:155 is added here when `n_bins` is given).

    mode "rank0": only rank `src` runs the pass, then `broadcast_module(flow, src)` - every replica ends up with what a single
    process given rank `src`'s batch would have (the parity semantics; also carries rank `src`'s random 1x1-conv weights).
    mode "global": parameters are broadcast first (so the activations agree), then every rank runs the pass on its own shard with the
    ActNorm moments merged across ranks: single-process semantics with the GLOBAL first batch."""
    if mode not in ("rank0", "global"):
        raise ValueError(f"data_dependent_init: mode {mode!r}")
    rank = dist.get_rank(group) if _multi(group) else src

    def run():
        was = flow.training
        flow.eval()
        x = batch if n_bins is None else batch + torch.rand_like(batch) / n_bins
        b = x.shape[0]
        ldj = torch.zeros(b, dtype=torch.float64, device=x.device)
        logp = torch.zeros(b, dtype=torch.float64, device=x.device)
        flow.transform(x, ldj, logp)
        flow.train(was)

    if mode == "rank0":
        if rank == src:
            run()
        broadcast_module(flow, src=src, group=group)
    else:
        broadcast_module(flow, src=src, group=group)
        with global_actnorm_init(group):
            run()


class BucketedGradAllReduce:
    """Data-parallel gradient averaging for the training step: parameters are grouped into flat buckets in
    REVERSE registration order (the order backward produces their gradients: last Glow level first), and a bucket's
    all-reduce(SUM) is launched asynchronously from a post-accumulate-grad hook as soon as its last gradient is ready, so
    the collectives run on RCCL's stream under the rest of backward.  Any number of backward passes may precede `finish()`: a
    bucket whose collective was launched by an earlier pass is reduced again from the accumulated gradients.  `finish()` (call
    it after the last loss.backward(), before the
    clips and the optimizer step, so the clips see the global-batch gradient like a single-process run would) waits,
    divides by the world size and scatters the averages back into the .grad tensors.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets default to 32 MiB so the 88 MB of L3K16 gradients go
    out as three collectives (one per Glow level) instead of ~600 latency-bound ones.
    """

    def __init__(self, params, bucket_bytes: int = 32 << 20, group=None):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._seen = [set() for _ in self.buckets]
        self._launched = set()
        self._inflight = []
        self._stale = set()
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _hook(self, p) -> None:
        if not self._active():
            return
        i = self._bucket_of[id(p)]
        seen = self._seen[i]
        if id(p) in seen:
            # a second backward before finish() (gradient accumulation, two losses): a collective already in flight carries the
            # first pass only.  Its result is dropped and the bucket is reduced again, whole, in finish().
            if i in self._launched:
                self._stale.add(i)
            return
        seen.add(id(p))
        if len(seen) == len(self.buckets[i]) and i not in self._launched:
            self._launch(i)

    def _launch(self, i: int) -> None:
        flat = torch.cat([p.grad.reshape(-1) for p in self.buckets[i]])
        dev = flat.device
        if dist.get_backend(self.group) == "gloo" and flat.is_cuda:
            flat = flat.cpu()                              # CPU rehearsal backend
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((i, flat, work, dev))
        self._launched.add(i)

    @torch.no_grad()
    def finish(self) -> None:
        if not self._active():
            return
        for i, seen in enumerate(self._seen):              # parameters that received no gradient this step: bucket not launched
            if i not in self._launched and (seen or any(p.grad is not None for p in self.buckets[i])):
                for p in self.buckets[i]:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._launch(i)
        if self._stale:                                    # buckets that saw another backward after their launch: reduce them again
            keep = []
            for item in self._inflight:
                if item[0] in self._stale:
                    item[2].wait()                         # drain the collective, drop its (partial) result
                else:
                    keep.append(item)
            self._inflight = keep
            for i in sorted(self._stale):                  # same order on every rank
                self._launch(i)
            self._stale.clear()
        world = dist.get_world_size(self.group)
        for i, flat, work, dev in self._inflight:
            work.wait()
            flat = flat.to(dev).div_(world)
            off = 0
            for p in self.buckets[i]:
                k = p.numel()
                p.grad.copy_(flat[off:off + k].view_as(p.grad))
                off += k
        self._inflight.clear()
        self._launched.clear()
        for seen in self._seen:
            seen.clear()

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []


import weakref

_FLAT_REDUCERS = weakref.WeakSet()     # live FlatGradAllReduce objects: the backward pass tells them which gradients are final (grads_ready)


def grads_ready(params, include: bool = True, producer_streams=()) -> None:
    """Called by a backward node of a chain model.  include=False (at the START of the node's backward): the gradients of every
    parameter AFTER `params` in forward order are final - the nodes that produce them ran earlier in the backward pass, and autograd
    runs AccumulateGrad nodes ahead of everything else that is ready.  include=True (at its END, when the node wrote its parameter
    gradients into the buffer itself and has joined its side stream): those of `params` too.  normalizing_flow.training.LevelFusedFn
    .backward calls both per Glow level.  Every live FlatGradAllReduce then starts the collectives of the buffer slices that lie
    wholly in the final part, while the backward pass of the earlier levels is still running.  No-op without a process group,
    during hipGraph capture, and for parameters of other optimizers.
    producer_streams: HIP streams other than the current one that wrote into the gradients being reported (the weight-gradient side
    stream of normalizing_flow.training).  The caller has joined them already (current stream waits on them); the reducer does not rely
    on that: before it hands a slice to the collective it records an event on each of them and makes the current stream wait for it, so
    the dependency RCCL's stream inherits from the current stream provably covers every producer."""
    if not _FLAT_REDUCERS:
        return
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return
    for r in list(_FLAT_REDUCERS):
        r._producers = tuple(producer_streams)
        try:
            r._suffix_ready(params, include)
        finally:
            r._producers = ()


class FlatGradAllReduce:
    """Gradient averaging on nfdp_hip.optim.FusedAdam's flat gradient buffer: the buffer is all-reduced IN PLACE in a few large slices
    (default 32 MiB: three collectives for the 88 MB of Glow L3 K16, sized for xGMI's per-link-bound rings) - no hooks, no gather /
    scatter copies, so the backward launches keep adding parameter gradients straight into the buffer (FusedAdam(direct_grads=True)).
    **Overlapped with backward**: the optimizer's parameters are laid out in forward order, so the backward pass finishes them from the
    end of the buffer; each Glow level reports when its gradients are final (`grads_ready`), and every slice that lies wholly behind
    that point is started at once - the last level's slice travels while levels 1 and 0 are still in backward.  `finish()`, after
    backward, starts what is left and waits for all of them (on the stream) before the clips and the Adam launch read the buffer.
    The buffer then holds the SUM over ranks; the 1/world of the average is folded into the optimizer's clip pass
    (`FusedAdam.grad_scale`, applied where `clip_value_sumsq_kernel` / `adam_step_kernel` read the gradient anyway) instead of a
    separate pass over the 88 MB.  From construction on, every `optimizer.step()` therefore expects a finish() before it; `remove()`
    detaches.  overlap=False: everything in finish().

    sync_state: at construction the optimizer's flat parameter buffer, both moment buffers and the (lr, step) pair take rank 0's
    values - replicas built from different seeds or initialised on different batches start identical."""

    def __init__(self, optimizer, bucket_bytes: int = 32 << 20, group=None, sync_state: bool = True, overlap: bool = True):
        self.opt, self.group = optimizer, group
        self.step_elems = max(1, bucket_bytes // 4)
        self.overlap = bool(overlap)
        self._works, self._sent_lo = [], None      # collectives in flight; lowest element already handed to a collective this step
        self.early_slices = 0                       # slices started before finish() in the last step (diagnostics / tests)
        self._early = 0
        self._producers = ()                        # streams named by the grads_ready call in progress
        if self._active():
            self.opt.grad_scale = 1.0 / dist.get_world_size(group)
            if sync_state:
                n = broadcast_tensors([self.opt.flat_param, self.opt.exp_avg, self.opt.exp_avg_sq, self.opt._hyper], src=0, group=group)
                if n:
                    from . import bump_param_epoch
                    bump_param_epoch()
            _FLAT_REDUCERS.add(self)

    def remove(self) -> None:
        self.opt.grad_scale = 1.0
        _FLAT_REDUCERS.discard(self)

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _bounds(self):
        """Slice boundaries, counted from the END of the buffer (the slice that is final first is a whole one)."""
        n = self.opt.flat_grad.numel()
        return list(range(n, 0, -self.step_elems)) + [0]                      # n, n - step, ..., 0

    @torch.no_grad()
    def _send(self, lo: int, hi: int) -> None:
        part = self.opt.flat_grad[lo:hi]
        if part.is_cuda:
            # explicit ordering: every stream that produced these gradients -> the current stream -> (torch's ProcessGroupNCCL makes its
            # own stream wait for the current stream at the call) the collective
            cur = torch.cuda.current_stream(part.device)
            for st in self._producers:
                if st is not None and st != cur:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    cur.wait_event(ev)
        if dist.get_backend(self.group) == "gloo" and part.is_cuda:      # CPU rehearsal backend
            host = part.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            part.copy_(host)
        else:
            self._works.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    @torch.no_grad()
    def _send_down_to(self, ready_lo: int) -> int:
        """Start the collectives of every not-yet-sent slice that lies wholly at or above element `ready_lo`; -> slices started."""
        b = self._bounds()
        sent_lo = b[0] if self._sent_lo is None else self._sent_lo
        started = 0
        for hi, lo in zip(b[:-1], b[1:]):
            if hi > sent_lo:
                continue                                                 # already travelling
            if lo < ready_lo:
                break
            self._send(lo, hi)
            sent_lo = lo
            started += 1
        self._sent_lo = sent_lo
        return started

    def _suffix_ready(self, params, include: bool = True) -> None:
        if not (self.overlap and self._active()):
            return
        flat = self.opt.flat_grad
        base, n = flat.data_ptr(), flat.numel()
        los, his = [], []
        for p in params:
            g = getattr(p, "grad", None)
            if g is None or g.data_ptr() < base:
                continue
            off = (g.data_ptr() - base) // 4
            if off < n:
                los.append(off)
                his.append(off + g.numel())
        if los:
            self._early += self._send_down_to(min(los) if include else max(his))

    @torch.no_grad()
    def finish(self) -> None:
        if not self._active():
            return
        self._send_down_to(0)
        for w in self._works:
            w.wait()
        self._works, self._sent_lo = [], None
        self.early_slices, self._early = self._early, 0
