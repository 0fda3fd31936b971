"""Data-parallel host logic: one process per GPU, batch sharded across ranks, no data-path collective.

The flow treats samples independently, so forward/NLL and sampling shard over the batch with no
exchange; the only collectives are the global bits/dim — all-reduce(SUM) of the fp64 pair (sum_b bpd_b, count) —
and MAX of a timing.  Backend "nccl" is RCCL over xGMI on MI355X; "gloo" drives the CPU tests.
"""
from typing import Tuple

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


class FlatGradAllReduce:
    """Gradient averaging on nfdp_hip.optim.FusedAdam's flat gradient buffer: after backward, `finish()` all-reduces the buffer IN PLACE
    in a few large slices (default 32 MiB: three collectives for the 88 MB of Glow L3 K16, sized for xGMI's per-link-bound rings)
    and scales by 1/world - no hooks, no gather / scatter copies, so the backward launches keep adding parameter gradients straight
    into the buffer (FusedAdam(direct_grads=True)).  The collectives are asynchronous; they are waited for (on the stream) at the end
    of finish(), before the clips and the Adam launch read the buffer."""

    def __init__(self, optimizer, bucket_bytes: int = 32 << 20, group=None):
        self.opt, self.group = optimizer, group
        self.step_elems = max(1, bucket_bytes // 4)

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    @torch.no_grad()
    def finish(self) -> None:
        if not self._active():
            return
        flat, world = self.opt.flat_grad, dist.get_world_size(self.group)
        gloo = dist.get_backend(self.group) == "gloo"
        works = []
        # last slices first: backward finishes the last Glow level's gradients first, and they sit at the end of the buffer
        for lo in reversed(range(0, flat.numel(), self.step_elems)):
            part = flat[lo:lo + self.step_elems]
            if gloo and part.is_cuda:                      # CPU rehearsal backend
                host = part.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                part.copy_(host)
            else:
                works.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        flat.mul_(1.0 / world)
