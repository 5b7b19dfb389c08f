"""Optimizer tail of the training step on the native path (host-side mirror of sc/optimizer.py:7-47
`configure_optimizer` and of the clip + step pair in sc/trainers/base.py:362-385).

`FusedAdamW` is a `torch.optim.Optimizer` (param_groups / state_dict / LR schedulers work unchanged; the state keys
are torch.optim.AdamW's `step`, `exp_avg`, `exp_avg_sq`, so `optimizer.pt` checkpoints are interchangeable) whose
`step(max_grad_norm=...)` runs two HIP kernels per parameter tensor instead of torch's ~7 foreach passes:
`cx_grad_sq_norm` (global gradient norm into one device double) and `cx_adamw_clip_step` (clip coefficient derived on
the device + decoupled-decay Adam in one pass).  On the engine's flat buffers that is 2 + 2 launches per step.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import _C


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")  # torch.optim.AdamW raises the same way
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._sq_norm: Optional[torch.Tensor] = None
        self.last_grad_norm: Optional[torch.Tensor] = None  # device scalar (what clip_grad_norm_ returns)

    def _check(self, p: torch.Tensor):
        if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.grad is None:
            raise RuntimeError("FusedAdamW needs contiguous fp32 device parameters")
        if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.is_sparse:
            raise RuntimeError("FusedAdamW needs contiguous dense fp32 gradients")
        if p.data_ptr() % 16 or p.grad.data_ptr() % 16:
            raise RuntimeError("FusedAdamW needs 16-byte aligned buffers")

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None):
        if closure is not None:
            raise NotImplementedError("closures are not used by the reference trainers")
        lib = _C.lib()
        stream = _C.cur_stream()
        live = [(g, p) for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not live:
            return None
        for _, p in live:
            self._check(p)
        clip = max_grad_norm is not None and max_grad_norm > 0
        sq_ptr = None
        if clip:
            dev = live[0][1].device
            if self._sq_norm is None or self._sq_norm.device != dev:
                self._sq_norm = torch.zeros(1, dtype=torch.float64, device=dev)
            self._sq_norm.zero_()
            for _, p in live:
                _C.check(lib.cx_grad_sq_norm(p.grad.data_ptr(), p.numel(), self._sq_norm.data_ptr(), stream),
                         "cx_grad_sq_norm")
            sq_ptr = self._sq_norm.data_ptr()
        for g, p in live:
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            b1, b2 = g["betas"]
            _C.check(lib.cx_adamw_clip_step(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                            st["exp_avg_sq"].data_ptr(), p.numel(), float(g["lr"]), float(b1), float(b2),
                                            float(g["eps"]), float(g["weight_decay"]), int(st["step"].item()), sq_ptr,
                                            float(max_grad_norm) if clip else 0.0, stream), "cx_adamw_clip_step")
        self.last_grad_norm = self._sq_norm.sqrt().float() if clip else None
        return None


def configure_optimizer(modules: Iterable[torch.nn.Module], args, fused: bool = True):
    """sc/optimizer.py:7-47 for towers of this package: `param_groups(weight_decay)` already splits the flat buffers by
    the reference's rule (squeeze().ndim < 2, "bias", LayerNorm, logit_scale -> no decay); trainable logit scales and
    other plain modules are sorted by the same rule."""
    decay, no_decay = [], []
    for m in modules:
        if hasattr(m, "param_groups"):
            gd, gn = m.param_groups(args.weight_decay)
            decay += [p for p in gd["params"] if p.requires_grad]
            no_decay += [p for p in gn["params"] if p.requires_grad]
            continue
        for name, p in m.named_parameters():
            if not p.requires_grad:
                continue
            if p.squeeze().ndim < 2 or "bias" in name or "logit_scale" in name:
                no_decay.append(p)
            else:
                decay.append(p)
    groups = [{"params": decay, "weight_decay": args.weight_decay, "lr": args.learning_rate},
              {"params": no_decay, "weight_decay": 0.0, "lr": args.learning_rate}]
    cls = FusedAdamW if fused else torch.optim.AdamW
    return cls(groups, betas=(args.adam_beta1, args.adam_beta2), eps=args.eps)


class EmaWeights:
    """Exponential moving average of a set of fp32 parameter tensors (sc/trainers/base.py:387-391: the trainer calls
    `self.model["ema"].update(model)` after every training step when an EMA model is configured).  The engines keep their
    parameters in two flat buffers per tower, so the whole update is one `cx_ema_update` pass per buffer; shadows live in fp32
    next to the masters.  `decay` is the per-update weight of the running average (the `beta` of ema_pytorch-style wrappers,
    which is the API the reference's call assumes)."""

    def __init__(self, params: Iterable[torch.Tensor], decay: float = 0.9999):
        if not 0.0 <= decay <= 1.0:
            raise ValueError(f"ema decay must be in [0, 1], got {decay}")
        self.decay = float(decay)
        self.params = [p for p in params]
        self.shadow = [p.detach().clone() for p in self.params]
        self.num_updates = 0

    @torch.no_grad()
    def update(self, *_):
        lib, stream = _C.lib(), _C.cur_stream()
        for sh, p in zip(self.shadow, self.params):
            if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() % 16 == 0 and sh.data_ptr() % 16 == 0:
                _C.check(lib.cx_ema_update(sh.data_ptr(), p.data_ptr(), p.numel(), self.decay, stream), "cx_ema_update")
            else:   # (small head parameters that are not 16-byte aligned views)
                sh.mul_(self.decay).add_(p.detach(), alpha=1.0 - self.decay)
        self.num_updates += 1

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow": [s.detach().cpu() for s in self.shadow]}

    def load_state_dict(self, sd):
        self.decay, self.num_updates = float(sd["decay"]), int(sd["num_updates"])
        for s, v in zip(self.shadow, sd["shadow"]):
            s.copy_(v)

    @torch.no_grad()
    def copy_to(self, params: Optional[Iterable[torch.Tensor]] = None):
        """Write the averaged weights into `params` (default: the tracked parameters themselves, e.g. before an export)."""
        for s, p in zip(self.shadow, self.params if params is None else params):
            p.copy_(s)
