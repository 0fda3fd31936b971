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
    """The Glow that encodes images into latent parts (and decodes samples) for the diffusion prior; optionally frozen.
    Same constructor arguments, attributes (`model`, `L`, `K`, `device`) and methods as reference normalizing_flow/__init__.py:16-106;
    the flow runs on the HIP path (`compute`: 'f32' | 'bf16' | None = library default)."""

    def __init__(self, model_dir: str, in_channel: int, L: int, K: int, learn_prior_mean_logs: bool, freeze_flow: bool,
                 compute: Optional[str] = None):
        super().__init__()
        flow = Glow(in_channel=in_channel, L=L, K=K, learn_prior_mean_logs=learn_prior_mean_logs, compute=compute)
        self.model = flow.to(flow.device)
        self.device, self.L, self.K = flow.device, L, K
        self.learn_prior_mean_logs = learn_prior_mean_logs
        self.freeze_flow = bool(freeze_flow)
        if model_dir:                                         # checkpoint written by save_model(): {"flow": state_dict, ...}
            state = torch.load(model_dir, map_location="cpu")["flow"]
            flow.load_state_dict(state)
        flow.requires_grad_(not self.freeze_flow)

    def is_frozen(self) -> bool:
        return self.freeze_flow

    def set_train_mode(self) -> None:
        """train() only for a trainable backbone; a frozen one stays in eval mode (reference :58-64)."""
        self.train(mode=not self.freeze_flow)

    def set_eval_mode(self) -> None:
        self.train(mode=False)

    def transform(self, x: Tensor, log_det_jac: Tensor) -> Tuple[list, Tensor]:
        """Encode: Glow.transform with logp=None, i.e. the Split priors are not evaluated (reference :70-83)."""
        latents, ldj, _unused = self.model.transform(x, log_det_jac, None)
        return latents, ldj

    def invert(self, latents: list) -> Tensor:
        return self.model.invert(latents)

    @torch.no_grad()
    def sample(self, latents: list, postprocess_func=None) -> Tensor:
        return self.model.sample(latents, postprocess_func=postprocess_func)
