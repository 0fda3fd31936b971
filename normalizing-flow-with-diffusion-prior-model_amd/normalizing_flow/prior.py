"""Gaussian priors — mirrors reference normalizing_flow/prior.py: IsotropicGaussian :11-50, GaussianPrior :53-99,
save_model :102-115.  Log-densities are per-sample fp64 reductions in one HIP launch; sampling takes the
N(0,1) noise from torch's device generator (or an injected `eps=` for reproducible parity tests).
"""
import os
import pickle
from typing import Optional

import numpy as np
import torch
from torch import Tensor

import nfdp_hip as K
from .base import Prior
from .utils import ZeroConv2d


class IsotropicGaussian(Prior):
    """Diagonal Gaussian with given mean / log-sd tensors (logical NCHW, same shape as x)  (reference prior.py:11-50)."""

    def __init__(self, mean: Tensor, logsd: Tensor):
        super().__init__()
        self.Log2PI = float(np.log(2 * np.pi))
        self.mean = mean
        self.logsd = logsd

    def _h_nhwc(self) -> Tensor:
        return torch.cat([K.to_nhwc(self.mean), K.to_nhwc(self.logsd)], dim=-1).contiguous()   # layout plumbing

    def compute_log_prob(self, x: Tensor) -> Tensor:
        logp = torch.zeros(x.size(0), device=x.device, dtype=torch.float64)
        K.gauss_logp(K.to_nhwc(x), self._h_nhwc(), 0, logp)
        return logp.float()

    def sample(self, shape: tuple = None, temperature: float = 1.0, eps: Optional[Tensor] = None) -> Tensor:
        h = self._h_nhwc()
        b, hh, ww, c2 = h.shape
        eps = torch.randn((b, hh, ww, c2 // 2), device=h.device) if eps is None else K.to_nhwc(eps).contiguous()
        out = torch.empty((b, hh, ww, c2 // 2), device=h.device, dtype=torch.float32)
        K.gauss_sample(h, 0, eps, out, temperature)
        return K.as_nchw(out)


class GaussianPrior(Prior):
    """Gaussian prior with learnable, spatially constant mean/log-sd  (reference prior.py:53-99).

    The reference pushes an all-zero tensor through ZeroConv2d(2C,2C): the result is bias*exp(3*logs) at every
    pixel, so only that [2C] vector is computed here (the conv weight is kept for state-dict parity)."""

    def __init__(self, in_channels: int, learn_prior_mean_logs: bool = True):
        super().__init__()
        self.in_channels = in_channels
        self.__conv = ZeroConv2d(2 * in_channels, 2 * in_channels, padding=(3 - 1) // 2) \
            if learn_prior_mean_logs else None
        if self.__conv is not None:
            self.__conv.to(self.device)

    def _h(self, device) -> Tensor:
        if self.__conv is None:
            return torch.zeros(2 * self.in_channels, device=device, dtype=torch.float32)
        return K.zeroconv_bias_only(self.__conv.bias, self.__conv.logs.view(-1))

    def accumulate_log_prob_nhwc(self, zh: Tensor, logp: Tensor) -> None:
        """logp[b] += log N(z_b)   (fp64, in place); one launch: the ZeroConv2d-on-zeros vector bias * exp(3 logs) is formed in it."""
        cv = self.__conv
        K.gauss_logp_prior(zh, None if cv is None else cv.bias, None if cv is None else cv.logs.view(-1), logp)

    @property
    def conv(self):
        return self.__conv

    def compute_log_prob(self, x: Tensor) -> Tensor:
        """(reference prior.py:70-83) -> fp32 [B]; differentiable when gradients are required (training.py)."""
        from . import training
        if torch.is_grad_enabled() and (x.requires_grad or training.grad_needed(self)):
            return training.gaussian_prior_log_prob(self, x)
        logp = torch.zeros(x.size(0), device=x.device, dtype=torch.float64)
        self.accumulate_log_prob_nhwc(K.to_nhwc(x), logp)
        return logp.float()

    def sample(self, shape: tuple = None, temperature: float = 1.0, eps: Optional[Tensor] = None) -> Tensor:
        """(reference prior.py:85-99); shape = (N, C, H, W)."""
        n, c, hh, ww = shape
        eps = torch.randn((n, hh, ww, c), device=self.device) if eps is None else K.to_nhwc(eps).contiguous()
        out = torch.empty((n, hh, ww, c), device=eps.device, dtype=torch.float32)
        K.gauss_sample(self._h(eps.device), 1, eps, out, temperature)
        return K.as_nchw(out)


def save_model(logger_obj, model, p_dist, optim, current_epoch, current_iteration, checkpoint_directory, text=""):
    """Checkpoint writer with the reference's file name and keys  (reference prior.py:102-115)."""
    logger_obj.info(f"Saving the model. {text}")
    if isinstance(p_dist, GaussianPrior):
        keys, prefix = ("flow", "prior_dist"), "gaussian"
    else:
        keys, prefix = ("nf_backbone", "diffusion_prior"), "diffusion"
    torch.save({
        keys[0]: model.state_dict(),
        keys[1]: p_dist.state_dict(),
        "optimizer": optim.state_dict(),
        "current_iter": current_iteration,
    }, os.path.join(checkpoint_directory, f"model_{prefix}_{str(current_epoch).zfill(3)}.pt"))


def load_model(model, p_dist, checkpoint_directory, epoch, optim=None, map_location="cpu"):
    """Reads a checkpoint written by `save_model` — or by the reference's `save_model` (prior.py:102-115; same file
    names `model_{gaussian|diffusion}_{epoch:03d}.pt`, same keys, same state-dict layouts) — into `model` / `p_dist`
    (and `optim` when given); returns the stored iteration counter.  This is what the reference's entry scripts do inline
    (run_baseline_experiment.py:112-114, normalizing_flow/__init__.py:55-57)."""
    if isinstance(p_dist, GaussianPrior):
        keys, prefix = ("flow", "prior_dist"), "gaussian"
    else:
        keys, prefix = ("nf_backbone", "diffusion_prior"), "diffusion"
    path = os.path.join(checkpoint_directory, f"model_{prefix}_{str(epoch).zfill(3)}.pt")
    checkpoint = torch.load(path, map_location=map_location)
    model.load_state_dict(checkpoint[keys[0]])
    p_dist.load_state_dict(checkpoint[keys[1]])
    if optim is not None and "optimizer" in checkpoint:
        optim.load_state_dict(checkpoint["optimizer"])
    return checkpoint.get("current_iter", 0)


def save_architecture(workdir, L, K, learn_prior_mean_logs):
    """`architecture.pkl` as the reference's baseline script writes it  (run_baseline_experiment.py:65-68)."""
    with open(os.path.join(workdir, "architecture.pkl"), "wb") as f:
        pickle.dump({"L": L, "K": K, "learn_prior_mean_logs": learn_prior_mean_logs}, f)


def load_architecture(workdir):
    """-> {"L", "K", "learn_prior_mean_logs"}  (read side: run_diffusion_prior_experiment.py)."""
    with open(os.path.join(workdir, "architecture.pkl"), "rb") as f:
        return pickle.load(f)
