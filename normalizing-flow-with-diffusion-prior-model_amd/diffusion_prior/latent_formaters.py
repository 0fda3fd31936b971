"""Latent formaters — mirrors reference diffusion_prior/latent_formaters.py: BaseFormater :13-86, IdentityFormater
:89-132, get_formater :247-265.  CatFormater (:135-244) is out of scope (unused by every recipe)."""
from abc import ABC, abstractmethod

import numpy as np
import torch.nn as nn

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


def get_formater(name: str, *args, **kwargs) -> BaseFormater:
    if name in ("identity", "IdentityFormater"):
        return IdentityFormater(*args, **kwargs)
    raise ValueError(f"Unknown latent formater {name!r} (CatFormater is out of scope of the MI355X hot path)")
