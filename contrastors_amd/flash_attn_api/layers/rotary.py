"""flash_attn.layers.rotary: K11 (sc/layers/embedding.py:10,618-745; sc/layers/attention.py:122-135).
`RotaryEmbedding` keeps the attribute set that contrastors' `VarLengthRotaryEmbedding` subclass touches
(`_cos_cached/_sin_cached/_cos_k_cached/_sin_k_cached/scale/interleaved/inv_freq/base/dim/pos_idx_in_fp32/
_seq_len_cached/_update_cos_sin_cache/_compute_inv_freq`)."""
from __future__ import annotations

from typing import Optional, Union

import torch

from ... import _C


def _tables(cos, sin):
    return cos.float().contiguous(), sin.float().contiguous()


class _ApplyRotary(torch.autograd.Function):
    """Rotate the first 64 dims of x:(T,H,64) (varlen, cu_seqlens) or (B,S,H,64) by in-sequence position."""

    @staticmethod
    def forward(ctx, x, cos, sin, cu_seqlens, max_seqlen, inplace):
        if x.dtype != torch.bfloat16 or x.shape[-1] != 64 or cos.shape[-1] != 32:
            raise NotImplementedError("rotary: bf16, head_dim 64, full rotary dim only")
        if x.dim() == 4:
            B, S, H, _ = x.shape
            cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=x.device)
            mx = S
        else:
            H = x.shape[1]
            cu, mx = cu_seqlens.to(torch.int32), int(max_seqlen)
            B = cu.numel() - 1
        out = x if inplace else x.clone()
        if not out.is_contiguous():
            raise NotImplementedError("rotary on a non-contiguous tensor")
        c, s = _tables(cos, sin)
        T = out.numel() // (H * 64)
        _C.check(_C.lib().cx_rotary_apply(out.data_ptr(), H * 64, cu.data_ptr(), c.data_ptr(), s.data_ptr(), B, H, T, mx,
                                          1, _C.cur_stream()), "rotary")
        ctx.save_for_backward(c, s, cu)
        ctx.meta = (B, H, T, mx)
        if inplace:
            ctx.mark_dirty(x)
        return out

    @staticmethod
    def backward(ctx, g):
        c, s, cu = ctx.saved_tensors
        B, H, T, mx = ctx.meta
        g = g.contiguous().clone()
        _C.check(_C.lib().cx_rotary_apply(g.data_ptr(), H * 64, cu.data_ptr(), c.data_ptr(), s.data_ptr(), B, H, T, mx,
                                          -1, _C.cur_stream()), "rotary bwd")
        return g, None, None, None, None, None


def apply_rotary_emb_func(x, cos, sin, interleaved=False, inplace=False, seqlen_offsets: Union[int, torch.Tensor] = 0,
                          cu_seqlens: Optional[torch.Tensor] = None, max_seqlen: Optional[int] = None):
    if interleaved:
        raise NotImplementedError("interleaved (GPT-J) rotary")
    if not (isinstance(seqlen_offsets, int) and seqlen_offsets == 0):
        raise NotImplementedError("seqlen_offsets (KV-cache inference)")
    return _ApplyRotary.apply(x, cos, sin, cu_seqlens, max_seqlen, inplace)


apply_rotary_emb = apply_rotary_emb_func


class _ApplyRotaryQKV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cos, sin):
        if qkv.dtype != torch.bfloat16 or qkv.dim() != 5 or qkv.shape[2] != 3 or qkv.shape[-1] != 64:
            raise NotImplementedError("apply_rotary_emb_qkv_: bf16 (B,S,3,H,64)")
        if not qkv.is_contiguous():
            raise NotImplementedError("apply_rotary_emb_qkv_ on a non-contiguous tensor")
        B, S, _, H, _ = qkv.shape
        cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=qkv.device)
        c, s = _tables(cos, sin)
        _C.check(_C.lib().cx_rotary_qkv_inplace(qkv.data_ptr(), cu.data_ptr(), c.data_ptr(), s.data_ptr(), B, H, B * S,
                                                S, 1, _C.cur_stream()), "rotary qkv")
        ctx.save_for_backward(c, s, cu)
        ctx.meta = (B, S, H)
        ctx.mark_dirty(qkv)
        return qkv

    @staticmethod
    def backward(ctx, g):
        c, s, cu = ctx.saved_tensors
        B, S, H = ctx.meta
        g = g.contiguous().clone()
        _C.check(_C.lib().cx_rotary_qkv_inplace(g.data_ptr(), cu.data_ptr(), c.data_ptr(), s.data_ptr(), B, H, B * S, S,
                                                -1, _C.cur_stream()), "rotary qkv bwd")
        return g, None, None


def apply_rotary_emb_qkv_(qkv, cos, sin, cos_k=None, sin_k=None, interleaved=False, seqlen_offsets=0):
    if interleaved or cos_k is not None:
        raise NotImplementedError("interleaved / xPos rotary")
    return _ApplyRotaryQKV.apply(qkv, cos, sin)


def apply_rotary_emb_kv_(*a, **k):
    raise NotImplementedError("apply_rotary_emb_kv_ (GQA / KV-cache) is out of the encoder scope")


class RotaryEmbedding(torch.nn.Module):
    def __init__(self, dim: int, base=10000.0, interleaved=False, scale_base=None, pos_idx_in_fp32=True, device=None):
        super().__init__()
        self.dim, self.base, self.pos_idx_in_fp32 = dim, float(base), pos_idx_in_fp32
        self.register_buffer("inv_freq", self._compute_inv_freq(device), persistent=False)
        self.interleaved, self.scale_base = interleaved, scale_base
        scale = ((torch.arange(0, dim, 2, device=device, dtype=torch.float32) + 0.4 * dim) / (1.4 * dim)
                 if scale_base is not None else None)
        self.register_buffer("scale", scale, persistent=False)
        self._seq_len_cached = 0
        self._cos_cached = self._sin_cached = self._cos_k_cached = self._sin_k_cached = None

    def _compute_inv_freq(self, device=None):
        return 1.0 / (self.base ** (torch.arange(0, self.dim, 2, device=device, dtype=torch.float32) / self.dim))

    def _update_cos_sin_cache(self, seqlen, device=None, dtype=None):
        if (seqlen > self._seq_len_cached or self._cos_cached is None or self._cos_cached.device != device
                or self._cos_cached.dtype != dtype):
            self._seq_len_cached = seqlen
            t = torch.arange(seqlen, device=device, dtype=torch.float32)
            inv_freq = self.inv_freq if self.inv_freq.dtype == torch.float32 else self._compute_inv_freq(device)
            freqs = torch.outer(t, inv_freq.to(device))
            self._cos_cached = torch.cos(freqs).to(dtype)
            self._sin_cached = torch.sin(freqs).to(dtype)

    def forward(self, qkv, kv=None, seqlen_offset=0, max_seqlen=None):
        if kv is not None or self.scale is not None:
            raise NotImplementedError("kv-packed / xPos rotary")
        seqlen = qkv.shape[1]
        self._update_cos_sin_cache(max_seqlen or seqlen, device=qkv.device, dtype=qkv.dtype)
        return apply_rotary_emb_qkv_(qkv, self._cos_cached, self._sin_cached, interleaved=self.interleaved)
