"""flash_attn.ops.activations.swiglu: K10 (sc/layers/mlp.py:4,75: `swiglu(gate, y)` = silu(gate) * y)."""
from __future__ import annotations

import torch

from ... import _C


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        shape = x.shape
        I = shape[-1]
        if I % 8:
            raise NotImplementedError("swiglu width must be a multiple of 8")
        yg = torch.cat([y.reshape(-1, I), x.reshape(-1, I)], dim=-1).to(torch.bfloat16).contiguous()  # [y | gate]
        T = yg.shape[0]
        act = torch.empty(T, I, dtype=torch.bfloat16, device=x.device)
        _C.check(_C.lib().cx_swiglu_fwd(yg.data_ptr(), act.data_ptr(), T, I, 0, _C.cur_stream()), "swiglu fwd")
        ctx.save_for_backward(yg)
        ctx.meta = (shape, x.dtype, y.dtype)
        return act.view(shape).to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        (yg,) = ctx.saved_tensors
        shape, xd, yd = ctx.meta
        I = shape[-1]
        T = yg.shape[0]
        d = dout.reshape(T, I).to(torch.bfloat16).contiguous()
        dyg = torch.empty_like(yg)
        _C.check(_C.lib().cx_swiglu_bwd(d.data_ptr(), yg.data_ptr(), dyg.data_ptr(), T, I, 0, _C.cur_stream()),
                 "swiglu bwd")
        return dyg[:, I:].reshape(shape).to(xd), dyg[:, :I].reshape(shape).to(yd)


def swiglu(x, y):
    return _SwiGLU.apply(x, y)
