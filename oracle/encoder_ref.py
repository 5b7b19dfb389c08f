"""fp32 CPU restatement of the reference encoder math (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows the reference's eager twin, the only path of the reference that runs without flash-attn:
  embeddings      sc/models/huggingface/modeling_hf_nomic_bert.py:980-1000  (word + type (+ pos))
  embedding LN    :1694-1696
  attention       :1345-1414  (Wqkv -> rotary :1074-1099,1185-1212 -> softmax(QK^T/sqrt(dh)) V -> out_proj)
  MLP             :1024-1028 (fc1 -> GELU(erf) -> fc2) / :1059-1071 (fc11, fc12, silu gate, fc2)
  post-norm block :1496-1514
  pooling         sc/models/biencoder/modeling_biencoder.py:79-90 (mean, no clamp), :44-49 (cls), :314-319 (normalize)
State-dict keys are the reference's.  Everything is plain differentiable torch, so gradients of
the oracle come from autograd over this restatement.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def rotary_tables(seqlen: int, dim: int, base: float, dtype=torch.float32, scaling_factor: Optional[float] = None,
                  max_trained_positions: int = 2048):
    """cos/sin (seqlen, dim/2): modeling_hf_nomic_bert.py:1148-1183 (fp32 positions and inverse frequencies).
    With `scaling_factor` (config.rotary_scaling_factor) the Dynamic-NTK rule of :1215-1235 applies on a fresh module:
    for seqlen > max_trained_positions the base grows by (f*seqlen/max - (f-1))^(dim/(dim-2))."""
    if scaling_factor and seqlen > max_trained_positions:
        base = base * ((scaling_factor * seqlen / max_trained_positions) - (scaling_factor - 1)) ** (dim / (dim - 2))
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    freqs = torch.outer(torch.arange(seqlen, dtype=torch.float32), inv_freq)
    return torch.cos(freqs).to(dtype), torch.sin(freqs).to(dtype)


def apply_rotary(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Non-interleaved (NeoX) rotation of (B,S,H,D) by position; modeling_hf_nomic_bert.py:1074-1099."""
    half = cos.shape[-1]
    S = x.shape[1]
    c = cos[:S].reshape(1, S, 1, half)
    s = sin[:S].reshape(1, S, 1, half)
    x1, x2 = x[..., :half], x[..., half: 2 * half]
    rot = torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)
    return torch.cat([rot, x[..., 2 * half:]], dim=-1)


def encoder_hidden_states(sd: Dict[str, torch.Tensor], cfg, input_ids: torch.Tensor,
                          attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """(B,S) ids -> (B,S,d) last hidden state.  `cfg` needs the NomicBertConfig field names."""
    B, S = input_ids.shape
    d, H = cfg.n_embd, cfg.n_head
    dh = d // H
    x = F.embedding(input_ids, sd["embeddings.word_embeddings.weight"])
    x = x + sd["embeddings.token_type_embeddings.weight"][0]
    if cfg.rotary_emb_fraction == 0:
        x = x + sd["embeddings.position_embeddings.weight"][:S].unsqueeze(0)
    x = F.layer_norm(x, (d,), sd["emb_ln.weight"], sd["emb_ln.bias"], cfg.layer_norm_epsilon)
    if attention_mask is None:
        attention_mask = torch.ones(B, S, dtype=torch.long, device=x.device)
    key_bias = torch.zeros(B, 1, 1, S, dtype=x.dtype, device=x.device)
    key_bias = key_bias.masked_fill(attention_mask.view(B, 1, 1, S) == 0, torch.finfo(x.dtype).min)
    cos = sin = None
    if cfg.rotary_emb_fraction > 0:
        cos, sin = rotary_tables(S, int(dh * cfg.rotary_emb_fraction), cfg.rotary_emb_base, x.dtype,
                                 getattr(cfg, "rotary_scaling_factor", None), getattr(cfg, "max_trained_positions", 2048))
        cos, sin = cos.to(x.device), sin.to(x.device)
    for l in range(cfg.n_layer):
        p = f"encoder.layers.{l}."
        qkv = F.linear(x, sd[p + "attn.Wqkv.weight"], sd.get(p + "attn.Wqkv.bias"))
        qkv = qkv.view(B, S, 3, H, dh)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        if cos is not None:
            q, k = apply_rotary(q, cos, sin), apply_rotary(k, cos, sin)
        scores = torch.einsum("bqhd,bkhd->bhqk", q, k) / math.sqrt(dh) + key_bias
        ctx = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(scores, dim=-1), v).reshape(B, S, d)
        attn = F.linear(ctx, sd[p + "attn.out_proj.weight"], sd.get(p + "attn.out_proj.bias"))
        x = F.layer_norm(attn + x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.layer_norm_epsilon)
        if cfg.activation_function == "swiglu":
            y = F.linear(x, sd[p + "mlp.fc11.weight"], sd.get(p + "mlp.fc11.bias"))
            gate = F.linear(x, sd[p + "mlp.fc12.weight"], sd.get(p + "mlp.fc12.bias"))
            m = y * F.silu(gate)
        else:
            m = F.gelu(F.linear(x, sd[p + "mlp.fc1.weight"], sd.get(p + "mlp.fc1.bias")))
        m = F.linear(m, sd[p + "mlp.fc2.weight"], sd.get(p + "mlp.fc2.bias"))
        x = F.layer_norm(m + x, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.layer_norm_epsilon)
    return x


def pool(hidden: torch.Tensor, attention_mask: Optional[torch.Tensor], pooling: str = "mean") -> torch.Tensor:
    if pooling == "cls":
        return hidden[:, 0]
    if attention_mask is None:
        return hidden.mean(dim=1)
    m = attention_mask.unsqueeze(-1).to(hidden.dtype)
    return (hidden * m).sum(dim=1) / attention_mask.sum(dim=1, keepdim=True).to(hidden.dtype)


def biencoder_embedding(sd, cfg, input_ids, attention_mask, pooling="mean", normalize=True, hamming=False):
    """BiEncoder.forward (modeling_biencoder.py:287-319) for the encoder trunks in scope."""
    h = encoder_hidden_states(sd, cfg, input_ids, attention_mask)
    e = pool(h, attention_mask, pooling)
    if hamming:
        e = F.layer_norm(e, (e.shape[-1],))
    return F.normalize(e, dim=-1) if normalize else e


def random_state_dict(cfg, seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic test weights with the reference's keys; std 0.02 matrices, LN gamma ~ 1, small biases."""
    g = torch.Generator().manual_seed(seed)
    d, I = cfg.n_embd, cfg.n_inner
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std  # noqa: E731
    sd = {"embeddings.word_embeddings.weight": rn(cfg.vocab_size, d),
          "embeddings.token_type_embeddings.weight": rn(cfg.type_vocab_size, d),
          "emb_ln.weight": 1 + rn(d, std=0.1), "emb_ln.bias": rn(d, std=0.1)}
    sd["embeddings.word_embeddings.weight"][cfg.pad_token_id] = 0
    if cfg.rotary_emb_fraction == 0:
        sd["embeddings.position_embeddings.weight"] = rn(cfg.max_position_embeddings, d)
    for l in range(cfg.n_layer):
        p = f"encoder.layers.{l}."
        sd[p + "attn.Wqkv.weight"] = rn(3 * d, d, std=0.05)
        sd[p + "attn.out_proj.weight"] = rn(d, d, std=0.05)
        if cfg.qkv_proj_bias:
            sd[p + "attn.Wqkv.bias"] = rn(3 * d, std=0.05)
            sd[p + "attn.out_proj.bias"] = rn(d, std=0.05)
        if cfg.activation_function == "swiglu":
            sd[p + "mlp.fc11.weight"] = rn(I, d, std=0.05)
            sd[p + "mlp.fc12.weight"] = rn(I, d, std=0.05)
        else:
            sd[p + "mlp.fc1.weight"] = rn(I, d, std=0.05)
            if cfg.mlp_fc1_bias:
                sd[p + "mlp.fc1.bias"] = rn(I, std=0.05)
        sd[p + "mlp.fc2.weight"] = rn(d, I, std=0.05)
        if cfg.mlp_fc2_bias:
            sd[p + "mlp.fc2.bias"] = rn(d, std=0.05)
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = 1 + rn(d, std=0.1)
            sd[p + n + ".bias"] = rn(d, std=0.1)
    return sd
