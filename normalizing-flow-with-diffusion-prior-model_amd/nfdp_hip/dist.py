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
    the collectives run on RCCL's stream under the rest of backward.  `finish()` (call it after loss.backward(), before the
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
        self._ready = [0] * len(self.buckets)
        self._inflight = []
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _hook(self, p) -> None:
        if not self._active():
            return
        i = self._bucket_of[id(p)]
        self._ready[i] += 1
        if self._ready[i] == len(self.buckets[i]):
            self._launch(i)

    def _launch(self, i: int) -> None:
        flat = torch.cat([p.grad.reshape(-1) for p in self.buckets[i]])
        dev = flat.device
        if dist.get_backend(self.group) == "gloo" and flat.is_cuda:
            flat = flat.cpu()                              # CPU rehearsal backend
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((i, flat, work, dev))

    @torch.no_grad()
    def finish(self) -> None:
        if not self._active():
            return
        for i, n in enumerate(self._ready):                # parameters that received no gradient this step: bucket not launched
            if 0 < n < len(self.buckets[i]) or (n == 0 and any(p.grad is not None for p in self.buckets[i])):
                for p in self.buckets[i]:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._launch(i)
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
        self._ready = [0] * len(self.buckets)

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
