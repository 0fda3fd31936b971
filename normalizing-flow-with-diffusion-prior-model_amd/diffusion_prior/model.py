"""Diffusion prior composition — mirrors reference diffusion_prior/model.py: DiffusionModel :9-76, DiffusionPrior :79-159.

One (Unet + GaussianDiffusion) per latent part of the flow, `auto_normalize=False` (reference :37-38).  The
reference keeps its per-part models in a plain Python list, so its state_dict() is empty and .to()/.train() do not
propagate; here they live in an nn.ModuleList under the attribute `priors` — extra keys
that the reference never wrote, nothing it did write is renamed — and `parameters(only_trainable)` keeps the
reference's generator semantics.
"""
import numpy as np
import torch
import torch.nn as nn

from normalizing_flow.base import default_device
from .gaussian_diffusion import Unet, GaussianDiffusion


class DiffusionModel(nn.Module):
    """A single diffusion model over one latent part  (reference model.py:9-76)."""

    def __init__(self, unet_kwargs, diffusion_kwargs, image_size, channels):
        super().__init__()
        network = Unet(**unet_kwargs, channels=channels)
        self.device = default_device()
        self.__diffusion = GaussianDiffusion(network, **diffusion_kwargs, image_size=image_size,
                                             auto_normalize=False).to(self.device)
        self.loss_type = self.__diffusion.loss_type

    @property
    def diffusion(self) -> GaussianDiffusion:
        return self.__diffusion

    def forward(self, x):
        return self.__diffusion(x)

    def sample_latent(self, n_samples: int, return_all_timesteps: bool = False):
        return self.__diffusion.sample(batch_size=n_samples, return_all_timesteps=return_all_timesteps)

    def sample_latent_given_start(self, start):
        """(reference model.py:57-70)"""
        d = self.__diffusion
        timesteps = torch.ones(start.size(0), dtype=torch.int64, device=start.device) * (d.num_timesteps - 1)
        x_t = d.q_sample(start, timesteps, noise=None)
        for t in reversed(range(0, d.num_timesteps)):
            x_t, _ = d.p_sample(x_t, t, None)
        return d.unnormalize(x_t)

    def interpolate_latent(self, latent_1, latent_2, lam):
        return self.__diffusion.interpolate(latent_1, latent_2, t=None, lam=lam)

    def calc_neg_log_likelihood_loop(self, x):
        return self.__diffusion.calc_neg_log_likelihood_loop(x)


class DiffusionPrior(nn.Module):
    """Models the flow's prior with one diffusion model per latent part  (reference model.py:79-159)."""

    def __init__(self, *, latent_formater, unet_kwargs, diffusion_kwargs, image_sizes, latent_channels):
        super().__init__()
        self.device = default_device()
        self.latent_formater = latent_formater
        self.priors = nn.ModuleList([DiffusionModel(unet_kwargs, diffusion_kwargs, image_size=image_sizes[i],
                                                    channels=latent_channels[i])
                                     for i in range(self.latent_formater.get_num_latent_parts())])
        self.loss_type = self.priors[0].loss_type

    def forward(self, latents: list) -> list:
        """List of per-part losses  (reference model.py:105-119)."""
        losses = []
        processed_latents = self.latent_formater.process_latents(latents)
        for i in range(0, len(processed_latents)):
            losses.append(self.priors[i](processed_latents[i]))
        return losses

    def sample_latents(self, n_samples: int, return_all_timesteps: bool = False) -> list:
        """(reference model.py:121-134)"""
        if not return_all_timesteps:
            return self.latent_formater.postprocess([prior.sample_latent(n_samples, False) for prior in self.priors])
        return [prior.sample_latent(n_samples, True) for prior in self.priors]

    @torch.no_grad()
    def sample_latents_given_start(self, latents: list) -> list:
        return [prior.sample_latent_given_start(latents[idx]) for idx, prior in enumerate(self.priors)]

    def evaluate_neg_log_likelihood(self, latents: list) -> list:
        neg_lls = []
        processed_latents = self.latent_formater.process_latents(latents)
        for i in range(0, len(processed_latents)):
            prior, latent = self.priors[i], processed_latents[i]
            neg_lls.append(prior.calc_neg_log_likelihood_loop(latent) / np.prod(latent.shape[1:]))
        return neg_lls

    def interpolate_latents(self, latents1, latents2, lam):
        return [prior.interpolate_latent(latents1[idx], latents2[idx], lam) for idx, prior in enumerate(self.priors)]

    def parameters(self, only_trainable=False):
        """Generator over all per-part parameters  (reference model.py:152-159)."""
        for prior in self.priors:
            for param in prior.parameters():
                if only_trainable and not param.requires_grad:
                    continue
                yield param
