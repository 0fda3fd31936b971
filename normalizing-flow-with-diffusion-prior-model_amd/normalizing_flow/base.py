"""Base classes of the flow API — mirrors the reference's normalizing_flow/base.py:9-83.

Differences that matter on MI355X: the device is `cuda:LOCAL_RANK` (one process per GPU) instead of the
reference's process-wide "cuda"; modules can be constructed on a CPU-only host (state-dict work), but every
compute method needs the GPU + the HIP library and raises otherwise.
"""
import os
from abc import ABC, abstractmethod
from typing import Any, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor


def default_device() -> torch.device:
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    return torch.device("cpu")


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


class Transform(nn.Module, ABC):
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


class Prior(nn.Module, ABC):
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
