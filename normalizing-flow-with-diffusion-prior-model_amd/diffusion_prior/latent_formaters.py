"""Latent formaters — mirrors reference diffusion_prior/latent_formaters.py: BaseFormater :13-86, IdentityFormater
:89-132, CatFormater :135-244, get_formater :247-265.  CatFormater's index reshuffles run on the squeeze / unsqueeze /
channel-copy HIP kernels (bit-exact moves)."""
import copy
from abc import ABC, abstractmethod

import numpy as np
import torch
import torch.nn as nn

import nfdp_hip as K

from normalizing_flow import calculate_output_shapes
from normalizing_flow.base import default_device


class BaseFormater(nn.Module, ABC):
    def __init__(self, L: int, in_channels: int, size: int):
        super().__init__()
        self.device = default_device()
        self.latent_dims = np.array(calculate_output_shapes(L=L, in_channels=in_channels, size=size))
        self.mins, self.maxs = None, None

    @abstractmethod
    def process_latents(self, latents: list) -> list:
        ...

    @abstractmethod
    def postprocess(self, latents: list) -> list:
        ...

    def get_num_latent_parts(self) -> int:
        return len(self.latent_dims)

    def standardize_latents(self, latents: list) -> list:
        return latents          # the reference's standardisation is commented out (:69-75)

    def inv_standardize_latents(self, latents: list) -> list:
        return latents


class IdentityFormater(BaseFormater):
    """(reference :89-132)"""

    def __init__(self, L: int, in_channels: int, size: int):
        super().__init__(L, in_channels, size)
        self.postprocessed_latent_shapes = self.latent_dims

    def process_latents(self, latents: list) -> list:
        assert len(latents) == len(self.latent_dims), "IdentityFormater expects L latent tensors from Diffusion prior."
        return self.standardize_latents(latents)

    def postprocess(self, latents: list) -> list:
        return self.inv_standardize_latents(latents)

    def get_input_shapes(self) -> list:
        return self.postprocessed_latent_shapes


class CatFormater(BaseFormater):
    """All latent parts brought to the spatial size of the middle part and concatenated along channels  (reference
    :135-244).  Part i is squeezed (target - i) times when it lies before the middle part (index (L-1)//2) and
    unsqueezed (i - target) times when after it; `postprocess` is the exact inverse.  The reference undoes the
    reshuffle by (un)squeezing the whole remaining channel block and peeling one part off per step (:205-229); because
    squeeze maps channel c to channels 4c..4c+3 that is the same index map as undoing each part on its own, which is
    what runs here (one kernel per squeeze level per part, reading channel slices in place).
    `get_input_shapes()` reports the reference's value — twice the middle part's channels (:151-153) — which is not the
    channel count of the concatenated tensor (e.g. 24 vs 48 for L=3 on 3x32x32); kept as is, see `processed_channels`."""

    def __init__(self, L: int, in_channels: int, size: int):
        super().__init__(L, in_channels, size)
        dim = copy.deepcopy(self.latent_dims[(len(self.latent_dims) - 1) // 2])
        dim[0] *= 2
        self.postprocessed_latent_shapes = [dim]
        self._target = (len(self.latent_dims) - 1) // 2
        # channels of part i once it is at the target resolution: c_i * 4^(target - i)
        self._proc_channels = [int(d[0]) * 4 ** (self._target - i) if i <= self._target
                               else int(d[0]) // 4 ** (i - self._target) for i, d in enumerate(self.latent_dims)]

    @property
    def processed_channels(self) -> int:
        """Channels of the tensor `process_latents` returns."""
        return sum(self._proc_channels)

    @staticmethod
    def _squeeze_n(xh, n):
        for _ in range(n):
            xh = K.squeeze(xh)
        return xh

    @staticmethod
    def _unsqueeze_n(xh, n):
        for _ in range(n):
            xh = K.unsqueeze(xh if xh.is_contiguous() else K.copy_channels(xh))
        return xh

    def process_latents(self, latents: list) -> list:
        """(reference :160-184)"""
        if len(latents) != len(self.latent_dims):
            raise ValueError(f"CatFormater expects {len(self.latent_dims)} latent parts, got {len(latents)}")
        parts = []
        for i, z in enumerate(latents):
            zh = K.to_nhwc(z)
            d = self._target - i
            parts.append(self._squeeze_n(zh, d) if d > 0 else self._unsqueeze_n(zh, -d))
        b, h, w, _ = parts[self._target].shape
        out = torch.empty((b, h, w, sum(p.shape[-1] for p in parts)), device=parts[0].device, dtype=torch.float32)
        c0 = 0
        for p in parts:
            K.copy_channels(p, out[..., c0:c0 + p.shape[-1]])
            c0 += p.shape[-1]
        return self.standardize_latents([K.as_nchw(out)])

    def postprocess(self, latents: list) -> list:
        """(reference :186-231)"""
        assert len(latents) == 1, "CatFormater expects a single latent tensor from Diffusion prior."
        zh = K.to_nhwc(self.inv_standardize_latents(latents)[0])
        if zh.shape[-1] != sum(self._proc_channels):
            raise ValueError(f"CatFormater.postprocess: expected {sum(self._proc_channels)} channels, got {zh.shape[-1]}")
        outs, c0 = [], 0
        for i, pc in enumerate(self._proc_channels):
            part = zh[..., c0:c0 + pc]
            c0 += pc
            d = self._target - i
            if d > 0:
                part = self._unsqueeze_n(part, d)
            elif d < 0:
                part = self._squeeze_n(part, -d)
            else:
                part = K.copy_channels(part)
            outs.append(K.as_nchw(part))
        return outs

    def get_num_latent_parts(self):
        return 1

    def get_input_shapes(self) -> list:
        return self.postprocessed_latent_shapes


def get_formater(name: str):
    """Formater class by name  (reference :247-265)."""
    if name == "IdentityFormater":
        return IdentityFormater
    if name == "CatFormater":
        return CatFormater
    raise ValueError("Invalid formater name")
