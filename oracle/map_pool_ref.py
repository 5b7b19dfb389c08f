"""TEST INFRASTRUCTURE (never imported by the product path).  fp32 torch restatement of the reference's attention-pooling
head, `pooling: map` -- sc/models/biencoder/modeling_biencoder.py:93-156 (MultiHeadAttentionPooling) over
sc/layers/attention.py:313-432 (FlashAttentionPooling) and sc/layers/mlp.py:8-34 (MLP).  Pinned to the reference's own class
by tests/golden/map_pool_tiny.npz (oracle/make_golden.py gen_map_pool; tests/test_oracle_golden.py).

State-dict keys are the reference's: attn.{Wq,Wkv,out_proj}.{weight,bias}, attn.latent, mlp.{fc1,fc2}.{weight,bias},
norm1.{weight,bias}."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def random_state_dict(d: int, inner: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    sd = {"attn.latent": r(1, 1, d) * d ** -0.5}
    for n, (o, i) in (("attn.Wq", (d, d)), ("attn.Wkv", (2 * d, d)), ("attn.out_proj", (d, d)), ("mlp.fc1", (inner, d)),
                      ("mlp.fc2", (d, inner))):
        sd[n + ".weight"] = r(o, i) * 0.05
        sd[n + ".bias"] = r(o) * 0.02
    sd["norm1.weight"] = 1 + 0.1 * r(d)
    sd["norm1.bias"] = 0.1 * r(d)
    return sd


def map_pool(sd, hidden: torch.Tensor, n_head: int, eps: float) -> torch.Tensor:
    """hidden (B, S, d) -> (B, d): token 0 of  hidden + mlp(norm1(attention(latent query; keys / values = hidden)))."""
    B, S, d = hidden.shape
    dh = d // n_head
    q = F.linear(sd["attn.latent"].expand(B, -1, -1), sd["attn.Wq.weight"], sd.get("attn.Wq.bias"))       # modeling: attention.py:378-379
    kv = F.linear(hidden, sd["attn.Wkv.weight"], sd.get("attn.Wkv.bias")).view(B, S, 2, n_head, dh)        # :381-383
    q = q.view(B, 1, n_head, dh)
    k, v = kv.unbind(2)
    att = torch.einsum("bshd,bthd->bhst", q, k) / dh ** 0.5                                                 # softmax_scale = 1 / norm_factor
    a = torch.einsum("bhst,bthd->bshd", att.softmax(-1), v).reshape(B, 1, d)
    a = F.linear(a, sd["attn.out_proj.weight"], sd.get("attn.out_proj.bias"))                              # :431
    normed = F.layer_norm(a, (d,), sd["norm1.weight"], sd["norm1.bias"], eps)                               # modeling_biencoder.py:149
    y = F.linear(F.gelu(F.linear(normed, sd["mlp.fc1.weight"], sd.get("mlp.fc1.bias"))), sd["mlp.fc2.weight"],
                 sd.get("mlp.fc2.bias"))                                                                    # mlp.py:30-34
    return (hidden + y)[:, 0]                                                                               # :150-156
