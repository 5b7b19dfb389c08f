"""CPU restatement (fp32 torch) of the MLM pretraining head + loss.  TEST INFRASTRUCTURE ONLY.

Follows the eager twin /root/reference/src/contrastors/models/huggingface/modeling_hf_nomic_bert.py
  prediction head transform :1606-1623 (Linear(bias=mlp_fc1_bias) -> SiLU for swiglu configs / GELU -> LayerNorm)
  decoder                   :1626-1637 (Linear to vocab, weight tied to the word embeddings :1718-1719)
  NomicBertForPreTraining   :1704-1765 (mean cross-entropy over positions whose label != -100)
on top of oracle.encoder_ref.encoder_hidden_states.  Pinned by tests/golden/mlm_*_tiny.npz, which the reference's own
NomicBertForPreTraining produced (oracle/make_golden.py::gen_mlm).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import encoder_ref

HEAD_KEYS = ("cls.predictions.transform.dense.weight", "cls.predictions.transform.dense.bias",
             "cls.predictions.transform.layer_norm.weight", "cls.predictions.transform.layer_norm.bias",
             "cls.predictions.decoder.bias")


def random_head_state_dict(cfg, seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    d, V = cfg.n_embd, cfg.vocab_size
    sd = {"cls.predictions.transform.dense.weight": torch.randn(d, d, generator=g) * 0.05,
          "cls.predictions.transform.layer_norm.weight": 1.0 + 0.1 * torch.randn(d, generator=g),
          "cls.predictions.transform.layer_norm.bias": 0.1 * torch.randn(d, generator=g)}
    if cfg.mlp_fc1_bias:
        sd["cls.predictions.transform.dense.bias"] = 0.1 * torch.randn(d, generator=g)
        sd["cls.predictions.decoder.bias"] = 0.1 * torch.randn(V, generator=g)
    return sd


def mlm_logits(trunk_sd, head_sd, cfg, input_ids, attention_mask) -> torch.Tensor:
    """(B, S, V) prediction scores."""
    h = encoder_ref.encoder_hidden_states(trunk_sd, cfg, input_ids, attention_mask)
    h = F.linear(h, head_sd["cls.predictions.transform.dense.weight"], head_sd.get("cls.predictions.transform.dense.bias"))
    if cfg.activation_function == "swiglu":
        h = F.silu(h)
    else:
        h = F.gelu(h, approximate="tanh" if cfg.activation_function in ("gelu_new", "gelu_fast", "gelu_pytorch_tanh") else "none")
    h = F.layer_norm(h, (cfg.n_embd,), head_sd["cls.predictions.transform.layer_norm.weight"],
                     head_sd["cls.predictions.transform.layer_norm.bias"], cfg.layer_norm_epsilon)
    return F.linear(h, trunk_sd["embeddings.word_embeddings.weight"], head_sd.get("cls.predictions.decoder.bias"))


def mlm_loss(trunk_sd, head_sd, cfg, input_ids, attention_mask, labels) -> torch.Tensor:
    logits = mlm_logits(trunk_sd, head_sd, cfg, input_ids, attention_mask)
    return F.cross_entropy(logits.flatten(0, 1), labels.flatten(), ignore_index=-100).float()
