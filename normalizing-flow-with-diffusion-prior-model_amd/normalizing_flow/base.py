"""Base classes of the flow API — mirrors the reference's normalizing_flow/base.py:9-83.

Differences that matter on MI355X: the device is `cuda:LOCAL_RANK` (one process per GPU, made current on first use)
instead of the reference's process-wide "cuda"; modules can be constructed on a CPU-only host (state-dict work), but every
compute method needs the GPU + the HIP library and raises otherwise.
"""
import os
from abc import ABC, abstractmethod
from typing import Any, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor


def default_device() -> torch.device:
    """The reference captures a process-wide "cuda" (= the current device) at import (base.py:18).  Here: one process per
    GPU.  Under a launcher (LOCAL_RANK set) the rank's GPU is made the CURRENT device the first time a module is built, so a
    reference-style script that never calls torch.cuda.set_device() still gets its tensors and its kernel launches on the
    same GPU (the C ABI queues on the current device's stream and nfdp_hip refuses tensors of any other device).  Without a
    launcher it is whatever device is current, like the reference."""
    if not torch.cuda.is_available():
        return torch.device("cpu")
    lr = os.environ.get("LOCAL_RANK")
    if lr is None:
        return torch.device("cuda", torch.cuda.current_device())
    idx = int(lr) % max(torch.cuda.device_count(), 1)
    if torch.cuda.current_device() != idx:
        torch.cuda.set_device(idx)
    return torch.device("cuda", idx)


def acc_in(t: Optional[Tensor]) -> Optional[Tensor]:
    """Working fp64 accumulator for a user accumulator (the reference's trainers pass fp64, its tests fp32)."""
    if t is None or t.dtype == torch.float64:
        return t
    return torch.zeros(t.shape, device=t.device, dtype=torch.float64)


def acc_out(orig: Optional[Tensor], work: Optional[Tensor]) -> Optional[Tensor]:
    """Fold the fp64 working delta back into the user's (in-place updated) accumulator."""
    if orig is not None and work is not orig:
        orig += work.to(orig.dtype)
    return orig


class _EpochOnApply:
    """Module._apply (.to / .cuda / .float ...) re-creates or rewrites parameter storage: the kernel-layout weight caches are
    keyed on (storage address, tensor version, parameter epoch), so bump the epoch and let them rebuild."""

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        import nfdp_hip
        nfdp_hip.bump_param_epoch()
        return out


class Transform(_EpochOnApply, nn.Module, ABC):
    """Invertible transformation f: x -> y  (reference base.py:9-46)."""

    def __init__(self):
        super().__init__()
        self.device = default_device()

    @abstractmethod
    def transform(self, x: Tensor, log_det_jac: Tensor, logp: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        ...

    @abstractmethod
    def invert(self, y: Tensor) -> Tensor:
        ...


class Prior(_EpochOnApply, nn.Module, ABC):
    """Prior distribution  (reference base.py:49-83)."""

    def __init__(self):
        super().__init__()
        self.device = default_device()

    @abstractmethod
    def sample(self, shape: Tuple[int, Any]) -> Tensor:
        ...

    @abstractmethod
    def compute_log_prob(self, x: Tensor) -> Tensor:
        ...
