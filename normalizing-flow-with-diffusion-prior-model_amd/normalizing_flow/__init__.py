"""MI355X-native drop-in for the reference's `normalizing_flow` package (hot-path symbols).

    from normalizing_flow import Glow, GaussianPrior, NFBackbone, ...

NFBackbone mirrors reference normalizing_flow/__init__.py:16-106.  The reference's trainers, data loaders and
metrics are out of scope and are not re-exported.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .base import Transform, Prior, default_device
from .glow import StepFlow, GlowBlock, Glow
from .prior import IsotropicGaussian, GaussianPrior, save_model, load_model, save_architecture, load_architecture
from .transforms import InvConv2d, ActNorm, AffineCoupling, Squeeze, Split, IdentityTransform
from .utils import (init_optimizer, preprocess_batch, postprocess_batch, initialize_with_zeros, calculate_output_shapes,
                    calculate_loss, data_dependent_nf_initialization, get_item, ZeroConv2d, Conv2dActNorm,
                    coupling_network)


class NFBackbone(nn.Module):
    """Frozen-or-trainable Glow used as the encoder/decoder of the diffusion prior
    (reference normalizing_flow/__init__.py:16-106)."""

    def __init__(self, model_dir: str, in_channel: int, L: int, K: int, learn_prior_mean_logs: bool, freeze_flow: bool,
                 compute: Optional[str] = None):
        super().__init__()
        self.L, self.K, self.learn_prior_mean_logs = L, K, learn_prior_mean_logs
        self.model = Glow(in_channel=in_channel, L=L, K=K, learn_prior_mean_logs=learn_prior_mean_logs, compute=compute)
        self.device = self.model.device
        self.model.to(self.device)
        self.freeze_flow = freeze_flow
        if model_dir:
            checkpoint = torch.load(model_dir, map_location=torch.device('cpu'))
            self.model.load_state_dict(checkpoint["flow"])
        for param in self.model.parameters():
            param.requires_grad = not self.freeze_flow

    def is_frozen(self):
        return self.freeze_flow

    def set_train_mode(self):
        if not self.is_frozen():
            self.train()
        else:
            self.eval()

    def set_eval_mode(self):
        self.eval()

    def transform(self, x: Tensor, log_det_jac: Tensor) -> Tuple[list, Tensor]:
        """Glow.transform with logp=None: the Split priors are skipped  (reference __init__.py:81)."""
        parts_of_latent_variable, log_det_jac, _ = self.model.transform(x, log_det_jac, None)
        return parts_of_latent_variable, log_det_jac

    def invert(self, latents: list) -> Tensor:
        return self.model.invert(latents)

    @torch.no_grad()
    def sample(self, latents: list, postprocess_func=None) -> Tensor:
        return self.model.sample(latents, postprocess_func)
