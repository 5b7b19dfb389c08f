"""Collectives of the InfoNCE path (host-side mirror of sc/distributed.py:5-29).

`gather_with_grad` is the one exchange step of the data-parallel loss: an all-gather of the
per-rank embeddings in RANK ORDER (labels depend on it, sc/loss.py:108-117) whose backward is a reduce-scatter(SUM)
of the gathered gradient.  On ROCm backend "nccl" is RCCL over xGMI; one fused buffer per call (not a list of W
tensors + torch.cat as the reference does) so each rank issues exactly one collective per direction.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _AllGatherCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t: torch.Tensor) -> torch.Tensor:
        W = dist.get_world_size()
        t = t.contiguous()
        out = torch.empty((W * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        ctx.n = t.shape[0]
        return out

    @staticmethod
    def backward(ctx, g: torch.Tensor) -> torch.Tensor:
        g = g.contiguous()
        rank = dist.get_rank()
        if dist.get_backend() == "nccl":
            out = torch.empty((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
            return out
        # gloo (CPU tests) has no reduce_scatter: all-reduce and keep our slice (same result)
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g[rank * ctx.n: (rank + 1) * ctx.n].contiguous()


def gather_with_grad(t: torch.Tensor) -> torch.Tensor:
    """sc/distributed.py:5-12 -- identity for world size 1, else autograd-aware rank-ordered concatenation."""
    if _world() == 1:
        return t
    if t.ndim == 0:
        t = t.unsqueeze(0)
    return _AllGatherCat.apply(t)


def gather(t: torch.Tensor) -> torch.Tensor:
    """sc/distributed.py:15-29 -- no-grad all-gather; the local slice keeps its own tensor (and autograd history)."""
    if _world() == 1:
        return t
    if t.ndim == 0:
        t = t.unsqueeze(0)
    gathered = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, t)
    gathered[dist.get_rank()] = t
    return torch.cat(gathered, dim=0)


def print_rank_zero(*args, **kwargs):
    if not dist.is_initialized() or dist.get_rank() == 0:
        print(*args, **kwargs)
