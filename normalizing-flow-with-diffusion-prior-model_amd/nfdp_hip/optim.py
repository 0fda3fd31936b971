"""Device-side fused optimizer: clip_grad_value_ + clip_grad_norm_ + Adam/AdamW in two HIP launches on flat buffers.

Drop-in for the three lines `clip_grad_value_(flow.parameters(), 1); clip_grad_norm_(flow.parameters(), 1);
optimizer.step()` of the reference's training loops (normalizing_flow/trainer.py:165-167, diffusion_prior/trainer.py):

    opt = FusedAdam(clipped=flow.parameters(), unclipped=prior.parameters(), lr=1e-3, clip_value=1, max_norm=1)
    ...
    opt.zero_grad(); loss.backward(); opt.step()

Parameters are re-pointed into one flat fp32 buffer and their .grad into another (autograd accumulates into the views
in place), so the step, the norm and an RCCL all-reduce each see ONE contiguous tensor instead of ~600.
"""
from typing import Iterable, Optional

import torch

from . import _call, _ptr, _stream, lib, bump_param_epoch


class FusedAdam:
    def __init__(self, clipped: Iterable[torch.nn.Parameter], unclipped: Iterable[torch.nn.Parameter] = (), lr: float = 1e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, adamw: bool = False,
                 clip_value: Optional[float] = None, max_norm: Optional[float] = None):
        self.clipped = [p for p in clipped if p.requires_grad]
        self.unclipped = [p for p in unclipped if p.requires_grad]
        self.params = self.clipped + self.unclipped
        if not self.params:
            raise ValueError("FusedAdam: no trainable parameters")
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise ValueError("FusedAdam: parameters must be fp32 tensors on one device")
        self.lr, self.betas, self.eps, self.weight_decay, self.adamw = lr, betas, eps, weight_decay, adamw
        self.clip_value, self.max_norm = clip_value, max_norm
        self.n = sum(p.numel() for p in self.params)
        self.n_clip = sum(p.numel() for p in self.clipped)
        self.flat_param = torch.empty(self.n, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self._ws = torch.empty(lib().nfdp_fused_adam_ws_bytes() if dev.type == "cuda" else 8192, device=dev, dtype=torch.uint8)
        # learning rate and step count live on the device ({lr, steps taken}): the launch reads them there, so a training step captured
        # in a hipGraph replays a correct Adam step every time (host scalars would freeze at their capture-time values)
        self._hyper = torch.tensor([float(lr), 0.0], device=dev, dtype=torch.float64)
        self._lr_uploaded = float(lr)
        self._host_steps = 0
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + k].view(p.shape)         # the parameter now lives in the flat buffer
                p.grad = self.flat_grad[off:off + k].view(p.shape)          # autograd accumulates into this view in place
                off += k

    def zero_grad(self, set_to_none: bool = False) -> None:
        """One memset; the .grad views are kept (set_to_none is accepted for signature compatibility and ignored)."""
        self.flat_grad.zero_()
        off = 0
        for p in self.params:                                              # re-attach views a caller may have dropped
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + k].view(p.shape)
            off += k

    @torch.no_grad()
    def step(self) -> None:
        if not self.flat_param.is_cuda:
            raise RuntimeError("FusedAdam.step: the HIP path needs GPU tensors (no CPU fallback)")
        if float(self.lr) != self._lr_uploaded:                # a scheduler changed .lr on the host: one 8-byte copy on the stream
            self._hyper[0:1].copy_(torch.tensor([float(self.lr)], dtype=torch.float64), non_blocking=True)
            self._lr_uploaded = float(self.lr)
        b1, b2 = self.betas
        _call("nfdp_fused_adam_step_dev_f32", _ptr(self.flat_param), _ptr(self.flat_grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
              self.n, self.n_clip, float(self.clip_value or 0.0), float(self.max_norm or 0.0), _ptr(self._hyper), float(b1), float(b2),
              float(self.eps), float(self.weight_decay), int(self.adamw), _ptr(self._ws), _ptr(self.grad_norm), _stream())
        self._host_steps += 1
        bump_param_epoch()          # parameters changed through an alias: invalidate the kernel-layout weight caches

    @property
    def step_count(self) -> int:
        """Steps taken, read back from the device counter (graph replays advance it without passing through step())."""
        return int(self._hyper[1].item())

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "lr": self.lr}

    def load_state_dict(self, sd) -> None:
        self._hyper[1] = float(int(sd["step"]))
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr = sd.get("lr", self.lr)
