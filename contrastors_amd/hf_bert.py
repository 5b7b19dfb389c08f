"""Hugging Face BERT <-> nomic-bert ("flash") checkpoint layouts (host-side mirror of sc/models/encoder/bert.py:11-366:
`bert_config_to_nomic_config`, `nomic_config_to_bert_config`, `remap_bert_state_dict`, `inv_remap_state_dict`).

Pure key/tensor bookkeeping on CPU tensors, no kernels.  Implemented as two rename tables (one per direction) plus the
three tensor-level edits the layouts differ by: q/k/v <-> Wqkv concatenation, vocabulary padding, and the decoder
bias alias.  tests/golden/hf_remap.npz holds what the reference's own functions return for the same inputs.
"""
from __future__ import annotations

import re
from collections import OrderedDict
from typing import Dict

import torch
import torch.nn.functional as F

from .nomic_bert import NomicBertConfig

# ---- rename tables (regex -> replacement), applied in order to every key ---------------------------------------------
_HF_TO_NOMIC = [
    (r"^roberta\.", "bert."),
    (r"^(?!bert\.|cls\.)", "bert."),                               # bare BertModel checkpoints
    (r"LayerNorm\.gamma$", "LayerNorm.weight"),
    (r"LayerNorm\.beta$", "LayerNorm.bias"),
    (r"^bert\.encoder\.layer\.", "bert.encoder.layers."),
    (r"^bert\.embeddings\.LayerNorm\.", "bert.emb_ln."),
    (r"^bert\.encoder\.layers\.(\d+)\.attention\.output\.LayerNorm\.(weight|bias)$", r"bert.encoder.layers.\1.norm1.\2"),
    (r"^bert\.encoder\.layers\.(\d+)\.output\.LayerNorm\.(weight|bias)$", r"bert.encoder.layers.\1.norm2.\2"),
    (r"^cls\.predictions\.transform\.LayerNorm\.(weight|bias)$", r"cls.predictions.transform.layer_norm.\1"),
    (r"^bert\.encoder\.layers\.(\d+)\.intermediate\.dense\.(weight|bias)$", r"bert.encoder.layers.\1.mlp.fc1.\2"),
    (r"^bert\.encoder\.layers\.(\d+)\.output\.dense\.(weight|bias)$", r"bert.encoder.layers.\1.mlp.fc2.\2"),
    (r"^bert\.encoder\.layers\.(\d+)\.attention\.output\.dense\.(weight|bias)$", r"bert.encoder.layers.\1.attn.out_proj.\2"),
    (r"^cls\.predictions\.bias$", "cls.predictions.decoder.bias"),
    (r"^bert\.lm_head\.bias$", "cls.predictions.decoder.bias"),
    (r"^bert\.lm_head\.dense\.(weight|bias)$", r"cls.predictions.transform.dense.\1"),
    (r"^bert\.lm_head\.layer_norm\.(weight|bias)$", r"cls.predictions.transform.layer_norm.\1"),
    (r"^bert\.lm_head\.decoder\.weight$", "cls.predictions.decoder.weight"),
]
_NOMIC_TO_HF = [
    (r"bert\.emb_ln\.", "bert.embeddings.LayerNorm."),
    (r"bert\.encoder\.layers\.(\d+)\.norm1\.(weight|bias)", r"bert.encoder.layers.\1.attention.output.LayerNorm.\2"),
    (r"bert\.encoder\.layers\.(\d+)\.norm2\.(weight|bias)", r"bert.encoder.layers.\1.output.LayerNorm.\2"),
    (r"cls\.predictions\.transform\.layer_norm\.(weight|bias)", r"cls.predictions.transform.LayerNorm.\1"),
    (r"bert\.encoder\.layers\.", "bert.encoder.layer."),
    (r"bert\.encoder\.layer\.(\d+)\.mlp\.fc1\.(weight|bias)", r"bert.encoder.layer.\1.intermediate.dense.\2"),
    (r"bert\.encoder\.layer\.(\d+)\.mlp\.fc2\.(weight|bias)", r"bert.encoder.layer.\1.output.dense.\2"),
    (r"bert\.encoder\.layer\.(\d+)\.attn\.out_proj\.(weight|bias)", r"bert.encoder.layer.\1.attention.output.dense.\2"),
]
_CLS_KEYS = ("cls.predictions.decoder.bias", "cls.predictions.transform.dense.weight",
             "cls.predictions.transform.dense.bias", "cls.predictions.transform.layer_norm.weight",
             "cls.predictions.transform.layer_norm.bias", "cls.predictions.decoder.weight")
_POOLER_AND_HEAD = ("bert.pooler.dense.weight", "bert.pooler.dense.bias", "bert.lm_head.bias", "bert.lm_head.dense.weight",
                    "bert.lm_head.dense.bias", "bert.lm_head.layer_norm.weight", "bert.lm_head.layer_norm.bias",
                    "bert.lm_head.decoder.weight")
_DROPPED = ("cls.seq_relationship.weight", "cls.seq_relationship.bias", "bert.embeddings.position_ids")


def _rename(sd, table) -> "OrderedDict[str, torch.Tensor]":
    rules = [(re.compile(p), r) for p, r in table]
    out = OrderedDict()
    for k, v in sd.items():
        for rx, rep in rules:
            k = rx.sub(rep, k)
        out[k] = v
    return out


def _n_layers(config) -> int:
    return getattr(config, "num_hidden_layers", None) or getattr(config, "n_layer")


def bert_config_to_nomic_config(bert_config) -> NomicBertConfig:
    """sc/models/encoder/bert.py:11-50 (the fields the native engine has; fused-op switches have no meaning here)."""
    g = lambda n, d=None: getattr(bert_config, n, d)  # noqa: E731
    return NomicBertConfig(
        vocab_size=bert_config.vocab_size, n_positions=bert_config.max_position_embeddings,
        max_position_embeddings=bert_config.max_position_embeddings, n_embd=bert_config.hidden_size,
        n_layer=bert_config.num_hidden_layers, n_head=bert_config.num_attention_heads,
        n_inner=bert_config.intermediate_size, activation_function=bert_config.hidden_act,
        resid_pdrop=bert_config.hidden_dropout_prob, embd_pdrop=bert_config.hidden_dropout_prob,
        attn_pdrop=bert_config.attention_probs_dropout_prob, layer_norm_epsilon=bert_config.layer_norm_eps,
        initializer_range=bert_config.initializer_range, prenorm=False, rotary_emb_fraction=g("rotary_emb_fraction", 0),
        qkv_proj_bias=g("qkv_proj_bias", True), rotary_emb_base=g("rotary_emb_base", 10_000),
        rotary_emb_interleaved=g("rotary_emb_interleaved", False), mlp_fc1_bias=g("mlp_fc1_bias", True),
        mlp_fc2_bias=g("mlp_fc2_bias", True), use_rms_norm=g("use_rms_norm", False), causal=False,
        type_vocab_size=bert_config.type_vocab_size, rotary_scaling_factor=g("rotary_scaling_factor", None),
        pad_token_id=bert_config.pad_token_id)


def nomic_config_to_bert_config(cfg):
    """sc/models/encoder/bert.py:53-72 -> transformers.BertConfig."""
    from transformers import BertConfig

    return BertConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.n_embd, num_hidden_layers=cfg.n_layer,
        num_attention_heads=cfg.n_head, intermediate_size=cfg.n_inner, hidden_act=cfg.activation_function,
        hidden_dropout_prob=cfg.resid_pdrop, attention_probs_dropout_prob=cfg.attn_pdrop,
        max_position_embeddings=cfg.n_positions, type_vocab_size=cfg.type_vocab_size,
        initializer_range=cfg.initializer_range, layer_norm_eps=cfg.layer_norm_epsilon, pad_token_id=cfg.pad_token_id,
        position_embedding_type="absolute", use_cache=True)


def remap_bert_state_dict(state_dict: Dict[str, torch.Tensor], config, remove_bert: bool = False,
                          remove_cls_weights: bool = False, add_pooling_layer: bool = False):
    """Hugging Face BERT / RoBERTa state dict -> the nomic-bert layout (sc/models/encoder/bert.py:75-261)."""
    sd = _rename(state_dict, _HF_TO_NOMIC[:12])
    # q / k / v Linear -> one Wqkv (rows q | k | v); the last layer keeps Wq + Wkv under `last_layer_subset`
    L = _n_layers(config)
    subset = getattr(config, "last_layer_subset", False)
    for l in range(L):
        pre = f"bert.encoder.layers.{l}.attention.self."
        if pre + "query.weight" not in sd:
            continue
        w = [sd.pop(pre + f"{n}.weight") for n in ("query", "key", "value")]
        b = [sd.pop(pre + f"{n}.bias") for n in ("query", "key", "value")]
        dst = f"bert.encoder.layers.{l}.attn."
        if subset and l == L - 1:
            sd[dst + "Wq.weight"], sd[dst + "Wkv.weight"] = w[0], torch.cat(w[1:], dim=0)
            sd[dst + "Wq.bias"], sd[dst + "Wkv.bias"] = b[0], torch.cat(b[1:], dim=0)
        else:
            sd[dst + "Wqkv.weight"], sd[dst + "Wqkv.bias"] = torch.cat(w, dim=0), torch.cat(b, dim=0)
    sd = _rename(sd, _HF_TO_NOMIC[11:12])  # keeps the reference's key order (out_proj renamed after the Wqkv insert)
    for k in _DROPPED:
        sd.pop(k, None)
    sd = _rename(sd, _HF_TO_NOMIC[12:])
    if remove_cls_weights:
        for k in _CLS_KEYS:
            sd.pop(k, None)
    # vocabulary padded to config.vocab_size; padded decoder-bias slots get -100 so they are never predicted
    if getattr(config, "pad_vocab_size_multiple", 1) > 1:
        we = sd["bert.embeddings.word_embeddings.weight"]
        sd["bert.embeddings.word_embeddings.weight"] = F.pad(we, (0, 0, 0, config.vocab_size - we.shape[0]))
        if not remove_cls_weights:
            dw = sd["cls.predictions.decoder.weight"]
            sd["cls.predictions.decoder.weight"] = F.pad(dw, (0, 0, 0, config.vocab_size - dw.shape[0]))
            if "cls.predictions.decoder.bias" in sd:
                db = sd["cls.predictions.decoder.bias"]
                sd["cls.predictions.decoder.bias"] = F.pad(db, (0, config.vocab_size - db.shape[0]), value=-100.0)
    if add_pooling_layer is False:
        for k in _POOLER_AND_HEAD:
            sd.pop(k, None)
    if getattr(config, "rotary_emb_fraction", 0.0) > 0.0:
        sd.pop("bert.embeddings.position_embeddings.weight", None)
    if remove_bert:
        sd = _rename(sd, [(r"^bert\.", "")])
    return sd


def inv_remap_state_dict(state_dict: Dict[str, torch.Tensor], config, require_cls: bool = True):
    """nomic-bert ("flash") layout -> Hugging Face BERT layout (sc/models/encoder/bert.py:264-366).  `state_dict` is
    consumed (keys popped), like the reference.  `require_cls=False` (not in the reference) converts a trunk-only
    checkpoint: the MLM-head entries are then optional."""
    sd = state_dict
    # the reference un-pads only for a BertConfig that carries pad_vocab_size_multiple / orig_vocab_size; a nomic
    # (GPT2-style) config goes through nomic_config_to_bert_config first, which has neither (:270-281)
    if not hasattr(config, "num_hidden_layers"):
        config = nomic_config_to_bert_config(config)
    if getattr(config, "pad_vocab_size_multiple", 1) > 1:
        n = config.orig_vocab_size
        for k in ("bert.embeddings.word_embeddings.weight", "cls.predictions.decoder.weight", "cls.predictions.decoder.bias"):
            if require_cls or k in sd:
                sd[k] = sd[k][:n]
    L = _n_layers(config)
    subset = getattr(config, "last_layer_subset", False)
    for l in range(L):
        src = f"bert.encoder.layers.{l}.attn."
        dst = f"bert.encoder.layers.{l}.attention.self."
        if subset and l == L - 1:
            wq, wkv = sd.pop(src + "Wq.weight"), sd.pop(src + "Wkv.weight")
            bq, bkv = sd.pop(src + "Wq.bias"), sd.pop(src + "Wkv.bias")
            ws, bs = (wq, *wkv.split(wkv.shape[0] // 2, dim=0)), (bq, *bkv.split(bkv.shape[0] // 2, dim=0))
        else:
            w, b = sd.pop(src + "Wqkv.weight"), sd.pop(src + "Wqkv.bias")
            ws, bs = w.split(w.shape[0] // 3, dim=0), b.split(b.shape[0] // 3, dim=0)
        for n, t in zip(("query", "key", "value"), ws):
            sd[dst + n + ".weight"] = t
        for n, t in zip(("query", "key", "value"), bs):
            sd[dst + n + ".bias"] = t
    sd = _rename(sd, _NOMIC_TO_HF)
    if require_cls or "cls.predictions.decoder.bias" in sd:
        sd["cls.predictions.bias"] = sd["cls.predictions.decoder.bias"]
    return sd


def load_hf_bert(engine, hf_state_dict: Dict[str, torch.Tensor], hf_config, strict: bool = True):
    """Load a Hugging Face BERT checkpoint into a NomicBertEngine trunk: what BiEncoder does for `bert-base-uncased`
    style models (sc/models/biencoder/modeling_biencoder.py:222-239 -> remap_bert_state_dict(remove_bert=True,
    remove_cls_weights=True, add_pooling_layer=False))."""
    sd = remap_bert_state_dict(OrderedDict(hf_state_dict), hf_config, remove_bert=True, remove_cls_weights=True,
                               add_pooling_layer=False)
    engine.load_reference_state_dict(sd, strict=strict)
    return engine


def export_hf_bert(engine) -> "OrderedDict[str, torch.Tensor]":
    """Trunk weights of a (GELU-MLP, biased) NomicBertEngine in the Hugging Face BERT layout (`bert.*` keys)."""
    if engine.config.gated or not engine.config.qkv_proj_bias:
        raise NotImplementedError("the HF BERT layout has q/k/v biases and a plain MLP; nomic-bert (SwiGLU, no biases) "
                                  "checkpoints keep the reference key layout (BiEncoder.save_pretrained)")
    sd = OrderedDict(("bert." + k, v.detach().cpu().clone()) for k, v in engine.reference_state_dict().items())
    return inv_remap_state_dict(sd, engine.config, require_cls=False)
