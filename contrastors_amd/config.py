"""YAML -> Config for the contrastive text-text recipes (host-side mirror of sc/config.py:8-241 and sc/read.py:5-11).

Field names and defaults follow the reference's pydantic models for the keys the north-star path reads
(`train_args`, `model_args`, `data_args`); the reference's own YAMLs (e.g. configs/train/contrastive_pretrain.yaml)
load unchanged -- unknown keys such as `use_fp8` are ignored exactly as there.  The two validators that guard this path
are kept: Matryoshka + GradCache is rejected (config.py:70-77) and eval_strategy needs eval_steps (:49-68).
"""
from __future__ import annotations

from typing import List, Optional, Union

import yaml
from pydantic import BaseModel, ConfigDict, model_validator


class TrainArgs(BaseModel):
    model_config = ConfigDict(extra="ignore", validate_assignment=True)
    num_epochs: int = 1
    num_train_steps: Optional[int] = None
    learning_rate: float = 2e-4
    weight_decay: float = 0.01
    eps: Optional[float] = 1e-8
    warmup_steps: Optional[int] = None
    warmup_pct: Optional[float] = None
    cooldown_steps: Optional[int] = None
    checkpoint: Optional[str] = None
    wandb: bool = False
    wandb_project_name: str = ""
    wandb_entity: str = ""
    wandb_run_name: Optional[str] = None
    wandb_group: Optional[str] = None
    log_grads_every: int = 100
    log_lr_every: int = 10
    save_every: Optional[int] = None
    eval_steps: Optional[int] = None
    eval_strategy: Optional[str] = None
    output_dir: Optional[str] = None
    gradient_accumulation_steps: Optional[int] = 1
    schedule_type: str = "cosine"
    max_grad_norm: float = 1.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    loss_fn: Optional[str] = "clip"
    grad_cache: Optional[bool] = None
    chunk_size: Optional[int] = None
    clamp_logits: Optional[bool] = True
    logit_max: Optional[float] = 100.0
    matryoshka_dims: Optional[List[int]] = None
    matryoshka_loss_weights: Optional[List[float]] = None
    profile: Optional[bool] = False
    # the reference YAMLs carry this key (contrastive_pretrain.yaml:24) without a TrainArgs field or any code behind it;
    # here it selects the fp8 matrix-core similarity GEMM of the fused InfoNCE (BASELINE configs[4])
    use_fp8: Optional[bool] = False
    # MI355X scheduling of the GradCache step (contrastors_amd.loss.GradCachePolicy; not reference keys).  The environment
    # variables CX_GRADCACHE_CHUNK / CX_GRADCACHE_RESIDENT / CX_EXCHANGE override these when set.
    gradcache_chunk: Union[int, str, None] = "auto"       # auto: chunk_size is a lower bound | exact | n
    gradcache_resident: Union[bool, str, None] = "auto"   # auto: keep pass 1's activations when they fit | true | false
    exchange: Optional[str] = "auto"                       # embedding exchange: auto (validated + timed at start-up) | rccl | oneshot
    exchange_timeout_s: float = 120.0                      # one-shot exchange: how long a GPU polls for a peer's signal before giving up
    overlap_grad_reduce: bool = True                       # per-block gradient all-reduce inside the step's last backward (DDP-style)
    # model_args.gradient_checkpointing is a memory knob sized for 80 GB parts: auto = the top blocks keep their activations
    # as far as the free HBM allows (bit-identical results, less re-forward) | n blocks | 0 = recompute every block
    checkpoint_keep_layers: Union[int, str, None] = "auto"

    @model_validator(mode="after")
    def _checks(self):
        from .policy import _parse_chunk, _parse_keep, _parse_resident   # (validate here, not at the first training step)

        _parse_chunk(self.gradcache_chunk, "train_args.gradcache_chunk")
        _parse_resident(self.gradcache_resident, "train_args.gradcache_resident")
        _parse_keep(self.checkpoint_keep_layers, "train_args.checkpoint_keep_layers")
        if self.exchange not in (None, "auto", "rccl", "oneshot"):
            raise ValueError(f"train_args.exchange must be auto, rccl or oneshot, got {self.exchange!r}")
        if self.use_fp8 and self.matryoshka_dims is not None:
            bad = [d for d in self.matryoshka_dims if d not in (256, 512, 768, 1024)]
            if bad:
                raise ValueError(f"use_fp8 covers similarity widths 256 / 512 / 768 / 1024; matryoshka_dims has {bad}")
        if self.eval_strategy is not None and self.eval_strategy not in ("steps", "epochs"):
            raise ValueError(f"Eval strategy {self.eval_strategy} not found in eval strategy registry")
        if self.eval_strategy == "steps" and self.eval_steps is None:
            raise ValueError("Eval steps must be set if eval strategy is set to steps")
        if self.matryoshka_dims is not None and self.grad_cache:
            raise ValueError("Matryoshka dims cannot be set if grad cache is set")
        return self


class DataArgs(BaseModel):
    model_config = ConfigDict(extra="ignore")
    shuffle: bool = False
    workers: int = 0
    batch_size: int = 16384
    seed: int = 42
    input_shards: Optional[str] = None
    download: Optional[bool] = False
    streaming: Optional[bool] = True
    process_one_shard: Optional[bool] = False
    weighted_sampling: Optional[bool] = False
    verbose: Optional[bool] = False
    sample_negatives: Optional[bool] = False
    query_max_length: Optional[int] = None
    document_max_length: Optional[int] = None
    mlm_prob: Optional[float] = None


class ModelArgs(BaseModel):
    model_config = ConfigDict(extra="ignore", protected_namespaces=())
    model_type: str = "encoder"
    model_name: Optional[str] = "nomic-ai/nomic-bert-2048"
    tokenizer_name: Optional[str] = "bert-base-uncased"
    seq_len: int = 2048
    logit_scale: Optional[float] = 1 / 0.07   # (None, as nomic_embed_vision_v1.5.yaml writes it, is the default: sc/config.py:193-197)
    trainable_logit_scale: bool = False
    pooling: str = "mean"
    nomic_encoder: bool = True
    add_prefix: bool = False
    num_negatives: Optional[int] = 7
    precomputed: Optional[bool] = False  # LiT: the batch carries `text_embs`, the (frozen) text tower is not run
    # architecture overrides of the MLM recipe (sc/config.py:152-170, configs/train/mlm.yaml); None = not given
    rotary_emb_fraction: Optional[float] = None
    rotary_emb_base: Optional[int] = 10_000
    pad_vocab_to_multiple_of: Optional[int] = None
    use_rms_norm: Optional[bool] = None
    activation_function: Optional[str] = "gelu"
    qkv_proj_bias: Optional[bool] = True
    mlp_fc1_bias: Optional[bool] = True
    mlp_fc2_bias: Optional[bool] = True
    attn_pdrop: Optional[float] = 0.0
    pretrained: Optional[bool] = True       # sc/config.py:161 (the reference default)
    checkpoint: Optional[str] = None         # sc/config.py:162: a BiEncoder.save_pretrained directory
    gradient_checkpointing: bool = False
    projection_dim: Optional[int] = None
    freeze: bool = False
    hamming: bool = False
    # sc/config.py:187 -> NomicBertModel.from_pretrained(..., resid_pdrop=) (modeling_biencoder.py:237): residual dropout of a
    # PRETRAINED nomic text trunk (None = the checkpoint's own value); served by the engine's Philox dropout
    resid_pdrop: Optional[float] = None
    ema: bool = False          # sc/config.py:179 + sc/trainers/base.py:387-391: an EMA copy of the weights, updated every step
    ema_decay: float = 0.9999  # (this path's key: the reference leaves the weighting as a TODO)
    patch_dropout: float = 0.0   # sc/config.py:180 -> PatchEmbedding's PatchDropout (image towers; sc/layers/embedding.py:415-418)
    # keys of the reference schema this path does not serve: accepted at their inert defaults, refused otherwise (a recipe
    # that sets them must not train as if it had not)
    num_experts: int = 0

    @model_validator(mode="after")
    def _model_type(self):
        if self.model_type not in ("encoder", "mlm", "glue", "locked_text", "image_text", "mmlm", "distill"):   # sc/config.py:198-203
            raise ValueError(f"Model type {self.model_type} not found in model registry")
        if not self.logit_scale:   # sc/config.py:193-196: `scale or 1 / 0.07`
            self.logit_scale = 1 / 0.07
        if not 0.0 <= self.ema_decay <= 1.0:
            raise ValueError(f"model_args.ema_decay must be in [0, 1], got {self.ema_decay}")
        if not 0.0 <= float(self.patch_dropout or 0.0) < 1.0:
            raise ValueError(f"model_args.patch_dropout must be in [0, 1), got {self.patch_dropout}")
        if self.num_experts and self.num_experts > 0:
            raise ValueError("model_args.num_experts > 0: mixture-of-experts trunks (megablocks) are out of scope")
        if self.resid_pdrop is not None and not (0.0 <= self.resid_pdrop < 1.0):
            raise ValueError(f"model_args.resid_pdrop must be in [0, 1), got {self.resid_pdrop}")
        return self


class Config(BaseModel):
    model_config = ConfigDict(extra="ignore", protected_namespaces=())
    train_args: TrainArgs
    data_args: DataArgs = DataArgs()
    model_args: ModelArgs = ModelArgs()
    # image-text recipes (sc/config.py:238-240): one ModelArgs per tower; `tower_model_args` (three towers) is not built
    text_model_args: Optional[ModelArgs] = None
    vision_model_args: Optional[ModelArgs] = None
    tower_model_args: Optional[ModelArgs] = None
    deepspeed: Optional[bool] = False
    deepspeed_config: Optional[dict] = None


def read_config(path: str) -> Config:
    with open(path) as f:
        return Config(**yaml.safe_load(f))
