"""flash_attn.flash_attn_interface: K1/K2 (sc/layers/attention.py:158-182,220-226)."""
from __future__ import annotations

import math

import torch

from .. import _C


def _scale(softmax_scale, d):
    if softmax_scale is None:
        return 1.0 / math.sqrt(d)
    return float(softmax_scale)  # contrastors passes a 0-dim tensor (attention.py:46,163)


def _draw_philox(device):
    """(seed, offset) from the device generator, advanced as a torch dropout op would (RandContext replays it)."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    off = gen.get_offset()
    gen.set_offset(off + 4)
    return gen.initial_seed() & (2**64 - 1), off


class _VarlenQKVPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_seqlen, scale, dropout_p=0.0):
        if qkv.dtype != torch.bfloat16 or qkv.dim() != 4 or qkv.shape[1] != 3 or qkv.shape[3] != 64:
            raise NotImplementedError("qkv must be bf16 (T,3,H,64)")
        qkv = qkv.contiguous()
        T, _, H, D = qkv.shape
        B = cu_seqlens.numel() - 1
        cu = cu_seqlens.to(torch.int32)
        out = torch.empty(T, H, D, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(H, T, dtype=torch.float32, device=qkv.device)
        rng = None
        if dropout_p:
            rng = _draw_philox(qkv.device)
            _C.check(_C.lib().cx_attn_varlen_dropout_fwd(qkv.data_ptr(), cu.data_ptr(), None, None, out.data_ptr(),
                                                         lse.data_ptr(), B, H, T, int(max_seqlen), scale, float(dropout_p),
                                                         rng[0], rng[1], 0, _C.cur_stream()), "attn fwd (dropout)")
        else:
            _C.check(_C.lib().cx_attn_varlen_fwd(qkv.data_ptr(), cu.data_ptr(), None, None, out.data_ptr(), lse.data_ptr(),
                                                 B, H, T, int(max_seqlen), scale, _C.cur_stream()), "attn fwd")
        ctx.save_for_backward(qkv, out, lse, cu)
        ctx.meta = (B, H, T, int(max_seqlen), scale, float(dropout_p), rng)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, cu = ctx.saved_tensors
        B, H, T, mx, scale, p_drop, rng = ctx.meta
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(H, T, dtype=torch.float32, device=qkv.device)
        if p_drop:
            _C.check(_C.lib().cx_attn_varlen_dropout_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                                         cu.data_ptr(), None, None, delta.data_ptr(), dqkv.data_ptr(), B, H, T,
                                                         mx, scale, p_drop, rng[0], rng[1], 0, _C.cur_stream()),
                     "attn bwd (dropout)")
        else:
            _C.check(_C.lib().cx_attn_varlen_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                                 cu.data_ptr(), None, None, delta.data_ptr(), dqkv.data_ptr(), B, H, T, mx,
                                                 scale, _C.cur_stream()), "attn bwd")
        return dqkv, None, None, None, None


def _check(dropout_p, causal, return_attn_probs, dropout_ok=False):
    if dropout_p and not dropout_ok:
        raise NotImplementedError("attention dropout > 0 is built for the qkv-packed self-attention forms only")
    if dropout_p and not 0.0 < dropout_p < 1.0:
        raise ValueError("dropout_p must be in [0, 1)")
    if causal:
        raise NotImplementedError("causal attention is out of the encoder hot-path scope")
    if return_attn_probs:
        raise NotImplementedError("return_attn_probs")


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                                     window_size=(-1, -1), alibi_slopes=None, deterministic=False,
                                     return_attn_probs=False):
    _check(dropout_p, causal, return_attn_probs, dropout_ok=True)
    return _VarlenQKVPacked.apply(qkv, cu_seqlens, max_seqlen, _scale(softmax_scale, qkv.shape[-1]), float(dropout_p))


def flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                              alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """(B,S,3,H,64) fixed-length form (the ViT path, attention.py:220-226): same kernel, cu_seqlens = arange * S."""
    _check(dropout_p, causal, return_attn_probs, dropout_ok=True)
    B, S = qkv.shape[:2]
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=qkv.device)
    out = _VarlenQKVPacked.apply(qkv.reshape(B * S, *qkv.shape[2:]), cu, S, _scale(softmax_scale, qkv.shape[-1]),
                                 float(dropout_p))
    return out.view(B, S, *out.shape[1:])


class _VarlenKVPacked(torch.autograd.Function):
    """K3: q (Tq,H,64) x kv (Tk,2,H,64) cross-attention per batch entry (cx_attn_varlen_kvpacked_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, q, kv, cu_q, cu_k, max_q, max_k, scale):
        if q.dtype != torch.bfloat16 or q.dim() != 3 or q.shape[2] != 64:
            raise NotImplementedError("q must be bf16 (Tq,H,64)")
        if kv.dtype != torch.bfloat16 or kv.dim() != 4 or kv.shape[1] != 2 or kv.shape[3] != 64:
            raise NotImplementedError("kv must be bf16 (Tk,2,H,64)")
        if kv.shape[2] != q.shape[1]:
            raise NotImplementedError("MQA / GQA (num_heads_kv != num_heads) is not built (the reference does not use it)")
        q, kv = q.contiguous(), kv.contiguous()
        Tq, H, D = q.shape
        B = cu_q.numel() - 1
        if cu_k.numel() != B + 1:
            raise ValueError("cu_seqlens_q and cu_seqlens_k must describe the same batch")
        cu_q, cu_k = cu_q.to(torch.int32), cu_k.to(torch.int32)
        out = torch.empty(Tq, H, D, dtype=q.dtype, device=q.device)
        lse = torch.empty(H, max(Tq, 1), dtype=torch.float32, device=q.device)
        _C.check(_C.lib().cx_attn_varlen_kvpacked_fwd(q.data_ptr(), kv.data_ptr(), cu_q.data_ptr(), cu_k.data_ptr(),
                                                      out.data_ptr(), lse.data_ptr(), B, H, Tq, int(max_q), int(max_k), scale,
                                                      _C.cur_stream()), "cross-attn fwd")
        ctx.save_for_backward(q, kv, out, lse, cu_q, cu_k)
        ctx.meta = (B, H, Tq, int(max_q), int(max_k), scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse, cu_q, cu_k = ctx.saved_tensors
        B, H, Tq, mq, mk, scale = ctx.meta
        dout = dout.contiguous()
        dq = torch.empty_like(q)
        dkv = torch.zeros_like(kv)   # (keys outside every cu_seqlens_k range, if any, get a zero gradient)
        delta = torch.empty(H, max(Tq, 1), dtype=torch.float32, device=q.device)
        _C.check(_C.lib().cx_attn_varlen_kvpacked_bwd(dout.data_ptr(), q.data_ptr(), kv.data_ptr(), out.data_ptr(),
                                                      lse.data_ptr(), cu_q.data_ptr(), cu_k.data_ptr(), delta.data_ptr(),
                                                      dq.data_ptr(), dkv.data_ptr(), B, H, Tq, mq, mk, scale,
                                                      _C.cur_stream()), "cross-attn bwd")
        return dq, dkv, None, None, None, None, None


def flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                                    softmax_scale=None, causal=False, window_size=(-1, -1), alibi_slopes=None,
                                    deterministic=False, return_attn_probs=False):
    """(Tq,H,64) x (Tk,2,H,64) -> (Tq,H,64): FlashAttentionPooling's varlen branch (sc/layers/attention.py:391-419)."""
    _check(dropout_p, causal, return_attn_probs)
    return _VarlenKVPacked.apply(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                 _scale(softmax_scale, q.shape[-1]))


def flash_attn_kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                             alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """(B,Sq,H,64) x (B,Sk,2,H,64) -> (B,Sq,H,64): FlashAttentionPooling's fixed-length branch (attention.py:420-428)."""
    _check(dropout_p, causal, return_attn_probs)
    B, Sq = q.shape[:2]
    Sk = kv.shape[1]
    cu_q = torch.arange(0, (B + 1) * Sq, Sq, dtype=torch.int32, device=q.device)
    cu_k = torch.arange(0, (B + 1) * Sk, Sk, dtype=torch.int32, device=q.device)
    out = _VarlenKVPacked.apply(q.reshape(B * Sq, *q.shape[2:]), kv.reshape(B * Sk, *kv.shape[2:]), cu_q, cu_k, Sq, Sk,
                                _scale(softmax_scale, q.shape[-1]))
    return out.view(B, Sq, *out.shape[1:])
