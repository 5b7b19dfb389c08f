"""flash_attn.ops.layer_norm: K5/K6 (sc/layers/block.py:309-319,422-431,453-462; modeling_nomic_bert.py:534)."""
from __future__ import annotations

import torch

from ... import _C


def _as(t, dtype):
    return t if t.dtype == dtype and t.is_contiguous() else t.to(dtype).contiguous()


def _philox_keep_mask(n: int, p: float, device) -> torch.Tensor:
    """Boolean keep-mask of n elements from the device generator's Philox stream (the generator advances as under a torch
    dropout, so torch.manual_seed and RandContext govern it): the kernel scales a vector of ones in place, what survives
    is kept.  n is rounded up to the kernel's group of 4."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    off = gen.get_offset()
    gen.set_offset(off + 4)
    n4 = (n + 3) // 4 * 4
    ones = torch.ones(n4, dtype=torch.bfloat16, device=device)
    _C.check(_C.lib().cx_dropout_scale(ones.data_ptr(), n4, float(p), gen.initial_seed() & (2**64 - 1), off, 0,
                                       _C.cur_stream()), "dropout mask")
    return (ones != 0)[:n]


class _DropoutAddLN(torch.autograd.Function):
    """dropout_add_layer_norm: z = dropout_p(x0) + residual (the Philox keep-mask is applied by
    the caller below when p > 0); statistics and normalisation in fp32;
    `out` in x0's dtype; z kept in fp32 when `residual_in_fp32` or the residual is fp32, else in x0's dtype.  Every
    operand keeps its dtype down to the kernel (cx_layernorm_fwd_mixed / _bwd_mixed): fp32 or bf16 (fp16 is cast to bf16)."""

    @staticmethod
    def forward(ctx, x0, residual, weight, bias, eps, prenorm, residual_in_fp32, rms=False):
        shape = x0.shape
        d = shape[-1]
        f32 = torch.float32
        xdt = f32 if x0.dtype == f32 else torch.bfloat16
        rdt = None if residual is None else (f32 if residual.dtype == f32 else torch.bfloat16)
        zdt = f32 if (residual_in_fp32 or rdt == f32 or xdt == f32) else torch.bfloat16
        x = _as(x0.reshape(-1, d), xdt)
        r = None if residual is None else _as(residual.reshape(-1, d), rdt)
        rows = x.shape[0]
        out = torch.empty(rows, d, dtype=xdt, device=x.device)
        z = torch.empty(rows, d, dtype=zdt, device=x.device)
        mean = torch.empty(rows, dtype=f32, device=x.device)
        rstd = torch.empty(rows, dtype=f32, device=x.device)
        w = weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        if b is None and not rms:
            raise ValueError("LayerNorm needs a bias")
        flags = (1 if xdt == f32 else 0) | (2 if rdt == f32 else 0) | (4 if xdt == f32 else 0) | (8 if zdt == f32 else 0) \
            | (16 if rms else 0)
        _C.check(_C.lib().cx_layernorm_fwd_mixed(x.data_ptr(), _C.ptr(r), w.data_ptr(), _C.ptr(b), out.data_ptr(),
                                                 z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, d, float(eps), flags,
                                                 _C.cur_stream()), "layernorm fwd")
        ctx.save_for_backward(z, w, mean, rstd)
        ctx.meta = (shape, x0.dtype, None if residual is None else residual.dtype, prenorm, weight.dtype, flags, xdt, rdt, zdt,
                    bias is not None)
        out = out.view(shape).to(x0.dtype)
        if prenorm:
            return out, z.view(shape)
        return out

    @staticmethod
    def backward(ctx, dout, dz_in=None):
        z, w, mean, rstd = ctx.saved_tensors
        shape, in_dtype, res_dtype, prenorm, wdtype, flags, xdt, rdt, zdt, has_bias = ctx.meta
        d = shape[-1]
        rows = z.shape[0]
        do = _as(dout.reshape(-1, d), xdt)
        dze = _as(dz_in.reshape(-1, d), zdt) if (prenorm and dz_in is not None) else None
        dx = torch.empty(rows, d, dtype=xdt, device=z.device)
        dr = None if rdt is None else torch.empty(rows, d, dtype=rdt, device=z.device)
        dg = torch.zeros(d, dtype=torch.float32, device=z.device)
        db = torch.zeros(d, dtype=torch.float32, device=z.device) if has_bias else None
        _C.check(_C.lib().cx_layernorm_bwd_mixed(do.data_ptr(), z.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                 _C.ptr(dze), dx.data_ptr(), _C.ptr(dr), dg.data_ptr(), _C.ptr(db), rows, d,
                                                 flags, _C.cur_stream()), "layernorm bwd")
        dx0 = dx.view(shape).to(in_dtype)
        dres = None if res_dtype is None else dr.view(shape).to(res_dtype)
        return dx0, dres, dg.to(wdtype), (db.to(wdtype) if has_bias else None), None, None, None, None


def dropout_add_layer_norm(x0, residual, weight, bias, dropout_p, epsilon, rowscale=None, layerscale=None,
                           prenorm=False, residual_in_fp32=False, return_dropout_mask=False, _rms=False):
    if rowscale is not None or layerscale is not None:
        raise NotImplementedError("rowscale / layerscale")
    keep = None
    if dropout_p:
        if not 0.0 < dropout_p < 1.0:
            raise ValueError("dropout_p must be in [0, 1)")
        # z = x0 * mask / (1 - p) + residual: the mask multiply stays in x0's dtype (autograd gives dx0 = dz * mask / (1 - p)),
        # the add + LayerNorm run in the fused kernel.  (The native towers fuse the mask into the LayerNorm kernel:
        # cx_dropout_add_layernorm_fwd; this op-by-op surface takes the three extra elementwise passes.)
        keep = _philox_keep_mask(x0.numel(), float(dropout_p), x0.device).view(x0.shape)
        x0 = x0 * keep.to(x0.dtype) * (1.0 / (1.0 - float(dropout_p)))
    out = _DropoutAddLN.apply(x0, residual, weight, bias, epsilon, prenorm, bool(residual_in_fp32), bool(_rms))
    if return_dropout_mask:
        mask = keep if keep is not None else torch.ones_like(x0, dtype=torch.bool)
        return (*out, mask) if isinstance(out, tuple) else (out, mask)
    return out


def layer_norm(x, weight, bias, epsilon):
    return _DropoutAddLN.apply(x, None, weight, bias, epsilon, False, False)


def dropout_add_layer_norm_parallel_residual(*a, **k):
    raise NotImplementedError("ParallelBlock (GPT-J style) is decoder-only: out of scope (SURVEY.md §2b K7)")


class DropoutAddLayerNorm(torch.nn.Module):
    def __init__(self, hidden_size, prenorm=False, p=0.0, eps=1e-5, residual_in_fp32=False, device=None, dtype=None):
        super().__init__()
        self.prenorm, self.p, self.eps, self.residual_in_fp32 = prenorm, p, eps, residual_in_fp32
        self.weight = torch.nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.bias = torch.nn.Parameter(torch.zeros(hidden_size, device=device, dtype=dtype))

    def forward(self, x0, residual=None):
        return dropout_add_layer_norm(x0, residual, self.weight, self.bias, self.p if self.training else 0.0, self.eps,
                                      prenorm=self.prenorm, residual_in_fp32=self.residual_in_fp32)
