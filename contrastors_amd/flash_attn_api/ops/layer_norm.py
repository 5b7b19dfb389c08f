"""flash_attn.ops.layer_norm: K5/K6 (sc/layers/block.py:309-319,422-431,453-462; modeling_nomic_bert.py:534)."""
from __future__ import annotations

import torch

from ... import _C


class _DropoutAddLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, residual, weight, bias, eps, prenorm):
        shape = x0.shape
        d = shape[-1]
        in_dtype = x0.dtype
        x = x0.reshape(-1, d).to(torch.bfloat16).contiguous()
        r = None if residual is None else residual.reshape(-1, d).to(torch.bfloat16).contiguous()
        rows = x.shape[0]
        out = torch.empty_like(x)
        z = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        w, b = weight.float().contiguous(), bias.float().contiguous()
        _C.check(_C.lib().cx_layernorm_fwd(x.data_ptr(), _C.ptr(r), w.data_ptr(), b.data_ptr(), out.data_ptr(),
                                           z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, d, float(eps),
                                           _C.cur_stream()), "layernorm fwd")
        ctx.save_for_backward(z, w, mean, rstd)
        ctx.meta = (shape, in_dtype, None if residual is None else residual.dtype, prenorm, weight.dtype)
        if prenorm:
            return out.view(shape), z.view(shape)
        return out.view(shape)

    @staticmethod
    def backward(ctx, dout, dz_in=None):
        z, w, mean, rstd = ctx.saved_tensors
        shape, in_dtype, res_dtype, prenorm, wdtype = ctx.meta
        d = shape[-1]
        rows = z.shape[0]
        do = dout.reshape(-1, d).to(torch.bfloat16).contiguous()
        dze = None
        if prenorm and dz_in is not None:
            dze = dz_in.reshape(-1, d).to(torch.bfloat16).contiguous()
        dz = torch.empty_like(z)
        dg = torch.zeros(d, dtype=torch.float32, device=z.device)
        db = torch.zeros(d, dtype=torch.float32, device=z.device)
        _C.check(_C.lib().cx_layernorm_bwd(do.data_ptr(), None, z.data_ptr(), w.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), _C.ptr(dze), dz.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                           None, 0, rows, d, _C.cur_stream()), "layernorm bwd")
        dx0 = dz.view(shape).to(in_dtype)
        dres = None if res_dtype is None else dz.view(shape).to(res_dtype)
        return dx0, dres, dg.to(wdtype), db.to(wdtype), None, None


def dropout_add_layer_norm(x0, residual, weight, bias, dropout_p, epsilon, rowscale=None, layerscale=None,
                           prenorm=False, residual_in_fp32=False, return_dropout_mask=False):
    if dropout_p and torch.is_grad_enabled():
        raise NotImplementedError("residual dropout > 0 is not implemented (BASELINE configs use 0)")
    if rowscale is not None or layerscale is not None or return_dropout_mask:
        raise NotImplementedError("rowscale / layerscale / return_dropout_mask")
    return _DropoutAddLN.apply(x0, residual, weight, bias, epsilon, prenorm)


def layer_norm(x, weight, bias, epsilon):
    return _DropoutAddLN.apply(x, None, weight, bias, epsilon, False)


def dropout_add_layer_norm_parallel_residual(*a, **k):
    raise NotImplementedError("ParallelBlock (GPT-J style) is decoder-only: out of scope (SURVEY.md §2b K7)")


class DropoutAddLayerNorm(torch.nn.Module):
    def __init__(self, hidden_size, prenorm=False, p=0.0, eps=1e-5, residual_in_fp32=False, device=None, dtype=None):
        super().__init__()
        self.prenorm, self.p, self.eps = prenorm, p, eps
        self.weight = torch.nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.bias = torch.nn.Parameter(torch.zeros(hidden_size, device=device, dtype=dtype))

    def forward(self, x0, residual=None):
        return dropout_add_layer_norm(x0, residual, self.weight, self.bias, self.p if self.training else 0.0, self.eps,
                                      prenorm=self.prenorm)
