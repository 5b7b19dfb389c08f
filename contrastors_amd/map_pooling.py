"""`pooling: map` -- multi-head attention pooling of the image tower (host-side mirror of
sc/models/biencoder/modeling_biencoder.py:93-156 MultiHeadAttentionPooling and sc/layers/attention.py:313-432
FlashAttentionPooling; the vision recipes use it: configs/train/nomic_embed_vision_v1.5.yaml:69).

    q   = Wq(latent)                       one learned query per image            (B, 1, H, 64)
    kv  = Wkv(h)                           keys / values of all tokens            (B, S, 2, H, 64)
    a   = out_proj(softmax(q k^T / 8) v)                                          (B, 1, d)
    emb = h[:, 0] + mlp(norm1(a))[:, 0]    the reference adds the head's output to the HIDDEN STATES and takes token 0
                                           (modeling_biencoder.py:150-156) -- reproduced as it is

Parameter names are the reference's (`attn.Wq`, `attn.Wkv`, `attn.latent`, `attn.out_proj`, `mlp.fc1`, `mlp.fc2`, `norm1`),
so a reference checkpoint's `selector.*` keys load unchanged.  The reference's masked (text) branch cannot run -- it expands
the latent over the unpadded token count and then un-pads it against a (B, S) mask -- so, like there, only unmasked
fixed-length inputs (the ViT tower) are served; `pooling: last` needs a decoder trunk's eos token and stays out of scope.

Compute: the three projections are the HIP bf16 MFMA GEMM (flash_attn_api FusedDense), the attention core is the K3
kv-packed kernel (cx_attn_varlen_kvpacked_fwd/_bwd) through `flash_attn_kvpacked_func`; LayerNorm / GELU of the (B, d)
head output run in torch on the device (B rows: nothing to fuse).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .flash_attn_api.flash_attn_interface import flash_attn_kvpacked_func
from .flash_attn_api.ops.fused_dense import FusedDense


class _PoolingAttention(torch.nn.Module):
    def __init__(self, d: int, n_head: int, bias: bool, device=None):
        super().__init__()
        if d != n_head * 64:
            raise NotImplementedError("attention pooling: head_dim 64 only (the K3 kernel)")
        self.n_head = n_head
        self.Wq = FusedDense(d, d, bias=bias, device=device)
        self.Wkv = FusedDense(d, 2 * d, bias=bias, device=device)
        self.out_proj = FusedDense(d, d, bias=bias, device=device)
        self.latent = torch.nn.Parameter(torch.zeros(1, 1, d, device=device))
        torch.nn.init.trunc_normal_(self.latent, std=d ** -0.5, a=-2 * d ** -0.5, b=2 * d ** -0.5)  # attention.py:353-354

    def forward(self, h: torch.Tensor) -> torch.Tensor:
        B, S, d = h.shape
        q = self.Wq(self.latent.expand(B, -1, -1)).view(B, 1, self.n_head, 64)
        kv = self.Wkv(h).view(B, S, 2, self.n_head, 64)
        a = flash_attn_kvpacked_func(q, kv, 0.0, softmax_scale=1.0 / 8.0)
        return self.out_proj(a.reshape(B, 1, d))


class _PoolingMLP(torch.nn.Module):
    def __init__(self, d: int, inner: int, bias1: bool, bias2: bool, device=None):
        super().__init__()
        self.fc1 = FusedDense(d, inner, bias=bias1, device=device)
        self.fc2 = FusedDense(inner, d, bias=bias2, device=device)

    def forward(self, x):
        y = self.fc1(x)
        return self.fc2(F.gelu(y.float()).to(y.dtype))


class MultiHeadAttentionPooling(torch.nn.Module):
    def __init__(self, config, device=None):
        super().__init__()
        act = getattr(config, "activation_function", "gelu")
        if act not in ("gelu", "gelu_new", "gelu_fast", "gelu_pytorch_tanh"):
            raise NotImplementedError(f"attention pooling with activation {act!r} (gated / quick_gelu heads are not built)")
        if getattr(config, "use_rms_norm", False):
            raise NotImplementedError("attention pooling with RMSNorm")
        d = config.n_embd
        self.attn = _PoolingAttention(d, config.n_head, bool(getattr(config, "qkv_proj_bias", True)), device=device)
        self.mlp = _PoolingMLP(d, config.n_inner, bool(getattr(config, "mlp_fc1_bias", True)),
                               bool(getattr(config, "mlp_fc2_bias", True)), device=device)
        self.norm1 = torch.nn.LayerNorm(d, eps=config.layer_norm_epsilon, device=device)

    def forward(self, hidden_states: torch.Tensor, input_ids=None, attention_mask=None) -> torch.Tensor:
        if attention_mask is not None:
            raise NotImplementedError("attention pooling of masked (text) inputs: the reference's own branch for it does not "
                                      "run (modeling_biencoder.py:134-148); pooling: map serves the image tower")
        a = self.attn(hidden_states)
        normed = self.norm1(a.float()).to(a.dtype)
        out = hidden_states[:, :1].float() + self.mlp(normed).float()
        return out[:, 0]
