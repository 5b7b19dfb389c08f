"""flash_attn.flash_attn_interface: K1/K2 (sc/layers/attention.py:158-182,220-226)."""
from __future__ import annotations

import math

import torch

from .. import _C


def _scale(softmax_scale, d):
    if softmax_scale is None:
        return 1.0 / math.sqrt(d)
    return float(softmax_scale)  # contrastors passes a 0-dim tensor (attention.py:46,163)


class _VarlenQKVPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_seqlen, scale):
        if qkv.dtype != torch.bfloat16 or qkv.dim() != 4 or qkv.shape[1] != 3 or qkv.shape[3] != 64:
            raise NotImplementedError("qkv must be bf16 (T,3,H,64)")
        qkv = qkv.contiguous()
        T, _, H, D = qkv.shape
        B = cu_seqlens.numel() - 1
        cu = cu_seqlens.to(torch.int32)
        out = torch.empty(T, H, D, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(H, T, dtype=torch.float32, device=qkv.device)
        _C.check(_C.lib().cx_attn_varlen_fwd(qkv.data_ptr(), cu.data_ptr(), None, None, out.data_ptr(), lse.data_ptr(),
                                             B, H, T, int(max_seqlen), scale, _C.cur_stream()), "attn fwd")
        ctx.save_for_backward(qkv, out, lse, cu)
        ctx.meta = (B, H, T, int(max_seqlen), scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, cu = ctx.saved_tensors
        B, H, T, mx, scale = ctx.meta
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(H, T, dtype=torch.float32, device=qkv.device)
        _C.check(_C.lib().cx_attn_varlen_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                             cu.data_ptr(), None, None, delta.data_ptr(), dqkv.data_ptr(), B, H, T, mx,
                                             scale, _C.cur_stream()), "attn bwd")
        return dqkv, None, None, None


def _check(dropout_p, causal, return_attn_probs):
    if dropout_p:
        raise NotImplementedError("attention dropout > 0 is not implemented (BASELINE configs use 0)")
    if causal:
        raise NotImplementedError("causal attention is out of the encoder hot-path scope")
    if return_attn_probs:
        raise NotImplementedError("return_attn_probs")


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                                     window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                                     return_attn_probs=False):
    _check(dropout_p, causal, return_attn_probs)
    return _VarlenQKVPacked.apply(qkv, cu_seqlens, max_seqlen, _scale(softmax_scale, qkv.shape[-1]))


def flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                              alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """(B,S,3,H,64) fixed-length form (the ViT path, attention.py:220-226): same kernel, cu_seqlens = arange * S."""
    _check(dropout_p, causal, return_attn_probs)
    B, S = qkv.shape[:2]
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=qkv.device)
    out = _VarlenQKVPacked.apply(qkv.reshape(B * S, *qkv.shape[2:]), cu, S, _scale(softmax_scale, qkv.shape[-1]))
    return out.view(B, S, *out.shape[1:])


def flash_attn_kvpacked_func(*a, **k):
    raise NotImplementedError("cross-attention (kv-packed) is only used by `pooling: map` -- out of round-1 scope")


def flash_attn_varlen_kvpacked_func(*a, **k):
    raise NotImplementedError("cross-attention (kv-packed) is only used by `pooling: map` -- out of round-1 scope")
