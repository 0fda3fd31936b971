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
