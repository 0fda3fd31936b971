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
    """weight_decay=None takes torch's default for the optimizer the reference's init_optimizer would build (utils.py:120-137):
    0 for 'adam', 0.01 for 'adamw'.

    direct_grads: "auto" | True | False - whether the backward launches may add parameter gradients straight into the flat gradient
    buffer, bypassing autograd's AccumulateGrad nodes (normalizing_flow.training._slot).  Anything hanging on those nodes then never
    runs - torch's DistributedDataParallel reducer does (its hooks live in C++ and cannot be detected) - so "auto" enables it only when
    no process group with more than one rank exists at construction; pass True under data parallelism only with a reducer that works
    on the flat buffer after backward (nfdp_hip.dist.FlatGradAllReduce), False with torch DDP even at world size 1."""

    def __init__(self, clipped: Iterable[torch.nn.Parameter], unclipped: Iterable[torch.nn.Parameter] = (), lr: float = 1e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: Optional[float] = None, adamw: bool = False,
                 clip_value: Optional[float] = None, max_norm: Optional[float] = None, direct_grads="auto"):
        if weight_decay is None:
            weight_decay = 0.01 if adamw else 0.0
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
        self.grad_scale = 1.0          # 1/world under nfdp_hip.dist.FlatGradAllReduce: flat_grad then holds the SUM over the ranks
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
        if direct_grads == "auto":
            import torch.distributed as dist
            direct_grads = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        self.direct_grads = bool(direct_grads)
        for p in self.params:
            p._nfdp_direct_grad = self.direct_grads

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
        _call("nfdp_fused_adam_step_dev_scaled_f32", _ptr(self.flat_param), _ptr(self.flat_grad), _ptr(self.exp_avg),
              _ptr(self.exp_avg_sq), self.n, self.n_clip, float(self.clip_value or 0.0), float(self.max_norm or 0.0), _ptr(self._hyper),
              float(b1), float(b2), float(self.eps), float(self.weight_decay), int(self.adamw), float(self.grad_scale), _ptr(self._ws),
              _ptr(self.grad_norm), _stream())
        self._host_steps += 1
        bump_param_epoch()          # parameters changed through an alias: invalidate the kernel-layout weight caches

    @property
    def step_count(self) -> int:
        """Steps taken, read back from the device counter (graph replays advance it without passing through step())."""
        return int(self._hyper[1].item())

    def _param_group(self):
        b1, b2 = self.betas
        return {"lr": self.lr, "betas": (b1, b2), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False,
                "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": bool(self.adamw),
                "params": list(range(len(self.params)))}

    def state_dict(self):
        """torch.optim.Adam / AdamW layout ({"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]}, parameter i =
        i-th of clipped + unclipped) so the reference's checkpoints (prior.py:102-115: "optimizer": optim.state_dict()) and
        `optimizer.load_state_dict` (normalizing_flow/trainer.py:115) interchange with it.  Before the first step the state is
        empty, like torch's."""
        n_steps = self.step_count
        state = {}
        if n_steps > 0:
            off = 0
            for i, p in enumerate(self.params):
                k = p.numel()
                state[i] = {"step": torch.tensor(float(n_steps)),
                            "exp_avg": self.exp_avg[off:off + k].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape).clone()}
                off += k
        return {"state": state, "param_groups": [self._param_group()]}

    def load_state_dict(self, sd) -> None:
        """Accepts torch's Adam/AdamW layout (any number of param groups; the moments are matched by parameter index; the learning
        rate, betas, eps and weight decay of the FIRST group apply - one flat launch has one set of hyper-parameters) and the flat
        layout earlier versions of this class wrote ({"step", "exp_avg", "exp_avg_sq", "lr"})."""
        if "param_groups" not in sd:                                       # legacy flat layout
            self._hyper[1] = float(int(sd["step"]))
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.lr = sd.get("lr", self.lr)
            return
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError(f"FusedAdam.load_state_dict: checkpoint has {len(ids)} parameters, this optimizer {len(self.params)}")
        state, steps, off = sd["state"], set(), 0
        with torch.no_grad():
            for i, p in zip(ids, self.params):
                k = p.numel()
                st = state.get(i)
                if st is None:                                             # parameter never stepped (torch keeps no state for it)
                    self.exp_avg[off:off + k].zero_()
                    self.exp_avg_sq[off:off + k].zero_()
                else:
                    if tuple(st["exp_avg"].shape) != tuple(p.shape):
                        raise ValueError(f"FusedAdam.load_state_dict: parameter {i} has shape {tuple(p.shape)}, "
                                         f"checkpoint {tuple(st['exp_avg'].shape)}")
                    self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(float(st["step"])))
                off += k
        if len(steps) > 1:
            raise ValueError(f"FusedAdam.load_state_dict: parameters carry different step counts {sorted(steps)}; "
                             "one flat launch has one bias correction")
        self._hyper[1] = float(steps.pop() if steps else 0)
        groups = sd["param_groups"]
        g0 = groups[0]
        # one flat launch = one plain Adam / AdamW: refuse what it cannot represent instead of loading it as something else
        for g in groups:
            if g.get("amsgrad", False) or g.get("maximize", False):
                raise ValueError("FusedAdam.load_state_dict: the checkpoint's optimizer uses amsgrad / maximize, which the fused "
                                 "launch does not implement")
            if "decoupled_weight_decay" in g and bool(g["decoupled_weight_decay"]) != bool(self.adamw):
                raise ValueError(f"FusedAdam.load_state_dict: checkpoint decoupled_weight_decay={g['decoupled_weight_decay']} but this "
                                 f"optimizer was built with adamw={self.adamw}")
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in g and k in g0 and tuple(np_flat(g[k])) != tuple(np_flat(g0[k])):
                    raise ValueError(f"FusedAdam.load_state_dict: param groups differ in {k} ({g0[k]} vs {g[k]}); one flat launch "
                                     "has one set of hyper-parameters")
        self.lr = float(g0.get("lr", self.lr))
        self.betas = tuple(g0.get("betas", self.betas))
        self.eps = float(g0.get("eps", self.eps))
        self.weight_decay = float(g0.get("weight_decay", self.weight_decay))

    @property
    def param_groups(self):
        """Read/write view for schedulers that set `group["lr"]` (normalizing_flow/trainer.py:119-120): assignments are picked up by
        the next step()."""
        return [_Group(self)]


def np_flat(v):
    return tuple(float(x) for x in v) if isinstance(v, (tuple, list)) else (float(v),)


class _Group(dict):
    """A param-group view: writes to lr / betas / eps / weight_decay reach the optimizer (next step()); the other keys describe what
    the fused launch is and cannot be changed."""
    _WRITABLE = {"lr": float, "eps": float, "weight_decay": float, "betas": lambda v: (float(v[0]), float(v[1]))}

    def __init__(self, opt):
        super().__init__(opt._param_group())
        super().__setitem__("params", opt.params)
        self._opt = opt

    def __setitem__(self, k, v):
        if hasattr(self, "_opt"):
            if k in self._WRITABLE:
                setattr(self._opt, k, self._WRITABLE[k](v))
            elif k not in self or self[k] != v:
                raise KeyError(f"FusedAdam param group: {k!r} is read-only (the fused launch is plain Adam/AdamW)")
        super().__setitem__(k, v)
