"""CPU restatement (fp32 torch) of the reference ViT image tower.  TEST INFRASTRUCTURE: only tests/, smoke() and the
bench's cpu_baseline leg may import this; the product path never does.

Follows, line by line in behaviour (not in code):
  * PatchEmbedding.forward         sc/layers/embedding.py:465-516  (rearrange "b c (h p1) (w p2) -> b h w (c p1 p2)",
                                    Linear, cls token prepended, + pos_embed over all P+1 positions)
  * Block.forward, pre-norm branch sc/layers/block.py:293-388      (residual = x + residual; LN1; attention;
                                    residual = attn + residual; LN2; MLP; returns (mlp_out, residual))
  * ViTModel.forward               sc/models/vit/vit.py:176-276    (blocks, then ln_f(mlp_out + residual))
  * MLP.forward                    sc/layers/mlp.py:30-34           (fc2(gelu_erf(fc1 x)))
  * FlashAttention (no rotary)     sc/layers/attention.py:217-229   softmax(q k^T / sqrt(head_dim)) v, non-causal
  * BiEncoder pooling              sc/models/biencoder/modeling_biencoder.py:44-49,79-90,317
The attention core itself lives in the third-party `flash-attn` package (unpinned, README.md:54); its published
algorithm (exact softmax attention) is what is restated.  Parity PINNED: tests/golden/vit_tiny.npz holds the outputs of
the reference's own ViTModel python (oracle/make_golden.py, attention core patched with the same exact softmax).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F


def random_state_dict(cfg, seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic test weights with the reference's keys (std 0.05 matrices, LN gamma ~ 1, small biases)."""
    g = torch.Generator().manual_seed(seed)
    d, I = cfg.n_embd, cfg.n_inner
    P = (cfg.img_size // cfg.patch_size) ** 2
    pd = cfg.num_channels * cfg.patch_size ** 2
    rn = lambda *s, std=0.05: torch.randn(*s, generator=g) * std  # noqa: E731
    sd = {"embeddings.cls_token": rn(1, 1, d, std=0.5), "embeddings.pos_embed": rn(1, P + 1, d, std=0.5),
          "embeddings.proj.weight": rn(d, pd), "embeddings.proj.bias": rn(d)}
    for l in range(cfg.n_layer):
        p = f"layers.{l}."
        sd[p + "attn.Wqkv.weight"], sd[p + "attn.Wqkv.bias"] = rn(3 * d, d), rn(3 * d)
        sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"] = rn(d, d), rn(d)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = rn(I, d), rn(I)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = rn(d, I), rn(d)
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
    sd["ln_f.weight"], sd["ln_f.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
    # the OpenAI-CLIP flavour (sc/models/vit/clip.py:14-58): pre-LayerNorm, no patch-embedding bias -- drawn LAST so that
    # the plain fixtures' weights do not move
    if getattr(cfg, "prepre_layernom", False):
        sd["prepre_layernom.weight"], sd["prepre_layernom.bias"] = 1 + rn(d, std=0.1), rn(d, std=0.1)
    if not getattr(cfg, "patch_embed_bias", True):
        del sd["embeddings.proj.bias"]
    return sd


def patchify(pixels: torch.Tensor, p: int) -> torch.Tensor:
    B, C, H, W = pixels.shape
    x = pixels.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (H // p) * (W // p), C * p * p)


def vit_hidden(sd: Dict[str, torch.Tensor], cfg, pixels: torch.Tensor, keep=None) -> torch.Tensor:
    """-> (B, P+1, d) output of ln_f.  keep (B, K) int64: PatchDropout (sc/layers/embedding.py:531-557) -- after the position
    embeddings every image keeps the [cls] token and its patch tokens keep[b], in that order: (B, K+1, d)."""
    d, H = cfg.n_embd, cfg.n_head
    eps = cfg.layer_norm_epsilon
    x = patchify(pixels.float(), cfg.patch_size) @ sd["embeddings.proj.weight"].T
    if "embeddings.proj.bias" in sd:
        x = x + sd["embeddings.proj.bias"]
    B = x.shape[0]
    x = torch.cat([sd["embeddings.cls_token"].expand(B, 1, d), x], 1) + sd["embeddings.pos_embed"]
    if keep is not None:
        keep = torch.as_tensor(keep, dtype=torch.long, device=x.device)
        x = torch.cat([x[:, :1], x[:, 1:][torch.arange(B, device=x.device)[:, None], keep]], 1)
    if "prepre_layernom.weight" in sd:   # sc/models/vit/vit.py:128-132,180
        x = F.layer_norm(x, (d,), sd["prepre_layernom.weight"], sd["prepre_layernom.bias"], eps)
    quick = getattr(cfg, "activation_function", "gelu") == "quick_gelu"   # sc/layers/activations.py:4-5
    act = (lambda t: t * torch.sigmoid(1.702 * t)) if quick else F.gelu
    hidden, residual = x, None
    for l in range(cfg.n_layer):
        p = f"layers.{l}."
        residual = hidden if residual is None else hidden + residual
        h = F.layer_norm(residual, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        qkv = (h @ sd[p + "attn.Wqkv.weight"].T + sd[p + "attn.Wqkv.bias"]).view(B, -1, 3, H, d // H)
        q, k, v = qkv.unbind(2)
        att = torch.einsum("bshd,bthd->bhst", q, k) / math.sqrt(d // H)
        ctx = torch.einsum("bhst,bthd->bshd", att.softmax(-1), v).reshape(B, -1, d)
        a = ctx @ sd[p + "attn.out_proj.weight"].T + sd[p + "attn.out_proj.bias"]
        residual = a + residual
        h = F.layer_norm(residual, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        y = act(h @ sd[p + "mlp.fc1.weight"].T + sd[p + "mlp.fc1.bias"])
        hidden = y @ sd[p + "mlp.fc2.weight"].T + sd[p + "mlp.fc2.bias"]
    return F.layer_norm(hidden + residual, (d,), sd["ln_f.weight"], sd["ln_f.bias"], eps)


def vit_embedding(sd, cfg, pixels, pooling: str = "cls", normalize: bool = True, keep=None) -> torch.Tensor:
    h = vit_hidden(sd, cfg, pixels, keep)
    e = h[:, 0] if pooling == "cls" else h.mean(1)
    return F.normalize(e, dim=-1) if normalize else e
