"""BiEncoder / LogitScale / DualEncoder on the native engine (host-side mirror of
sc/models/biencoder/modeling_biencoder.py:30-41,44-49,79-90,282-319, configuration_biencoder.py and
sc/models/dual_encoder/modeling_dual_encoder.py:36-68).

`BiEncoder(config).forward(input_ids, attention_mask, normalize)` -> {"embedding": (B,d) fp32}: same call and
result contract as the reference; trunk + pooling + L2-normalise are one native call (NomicBertEngine).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Union

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .distributed import gather_with_grad
from .nomic_bert import NomicBertConfig, NomicBertEngine, VarlenBatch, _EncodeFn
from .vit import ViTConfig, ViTEngine, _VitEncodeFn, _VitHiddenFn


@dataclass
class BiEncoderConfig:
    """sc/models/biencoder/configuration_biencoder.py:4-31 (same field names and defaults)."""

    model_name: str = "nomic-ai/nomic-bert-2048"
    projection_dim: Optional[int] = None
    logit_scale: float = 1 / 0.07
    use_fused_kernels: bool = True
    pooling: str = "last"
    nomic_encoder: bool = False
    freeze: bool = False
    trainable_logit_scale: bool = False
    hamming: bool = False
    pretrained: bool = False
    gradient_checkpointing: bool = False
    # (not a reference field) selective checkpointing on a 288 GB part: how many blocks keep their activations although
    # gradient_checkpointing is on -- "auto" = as many as the free HBM takes, 0 = the reference's behaviour (all recomputed)
    checkpoint_keep_layers: Union[int, str] = "auto"
    # model_args.resid_pdrop (sc/config.py:187; modeling_biencoder.py:237 hands it to the nomic text trunk): None = as configured
    resid_pdrop: Optional[float] = None
    # model_args.patch_dropout (sc/config.py:180; modeling_biencoder.py:174,187 sets it on the image trunk's config): fraction of
    # the patch tokens an image tower drops per image in training
    patch_dropout: float = 0.0
    encoder: bool = True
    seq_len: int = 2048
    trunk_config: Optional[object] = None  # NomicBertConfig or ViTConfig: no hub access, the architecture is explicit


class LogitScale(torch.nn.Module):
    """x * exp(log_scale) (sc/models/biencoder/modeling_biencoder.py:30-41)."""

    def __init__(self, config):
        super().__init__()
        self.logit_scale = torch.nn.Parameter(torch.ones([]) * np.log(config.logit_scale),
                                              requires_grad=config.trainable_logit_scale)
        # host copy of a frozen scale: lets the fused loss avoid a device sync per step
        self._const_scale = None if config.trainable_logit_scale else float(config.logit_scale)

    def forward(self, x):
        return x * self.logit_scale.exp()

    def __repr__(self):
        return f"LogitScale(logit_scale={self.logit_scale.exp().item()}, trainable={self.logit_scale.requires_grad})"


def _default_trunk_config(name: str) -> NomicBertConfig:
    if "nomic" in name:
        return NomicBertConfig.nomic_bert_2048()
    if "bert-base" in name:
        # a recipe that names the model gets what the reference builds from its hub config: dropout 0.1 included
        # (sc/models/encoder/bert.py:19-21); an explicit BiEncoderConfig.trunk_config is taken as given
        return NomicBertConfig.bert_base_uncased(hf_dropout=True)
    if "vit-base-patch16-224" in name or "vit_base_patch16_224" in name:
        return ViTConfig.vit_base_patch16_224()
    raise ValueError(f"no offline architecture table entry for {name!r}; pass BiEncoderConfig.trunk_config")


def config_json(config: "BiEncoderConfig", trunk_config) -> dict:
    """What save_pretrained writes as config.json: the reference's BiEncoderConfig fields under their own names (its
    PretrainedConfig.from_pretrained reads them back: tests/test_host_cpu.py), plus the explicit trunk architecture this
    build needs because it cannot ask a hub for it."""
    import dataclasses

    cfg = {k: v for k, v in dataclasses.asdict(config).items() if k != "trunk_config"}
    cfg["trunk_config"] = dataclasses.asdict(trunk_config)
    cfg["trunk_type"] = type(trunk_config).__name__
    return cfg


def trunk_config_with_overrides(config: BiEncoderConfig, trunk_cfg):
    """The architecture the tower is built with: the named / given trunk configuration with the overrides the reference's
    BiEncoder applies on top (modeling_biencoder.py:222-240: `resid_pdrop` of a text trunk)."""
    rp = getattr(config, "resid_pdrop", None)
    if rp is not None and isinstance(trunk_cfg, NomicBertConfig):
        import dataclasses

        trunk_cfg = dataclasses.replace(trunk_cfg, resid_pdrop=float(rp))
    pd = float(getattr(config, "patch_dropout", 0.0) or 0.0)
    if pd > 0.0 and hasattr(trunk_cfg, "patch_dropout"):   # image towers only (modeling_biencoder.py:174,187)
        import dataclasses

        trunk_cfg = dataclasses.replace(trunk_cfg, patch_dropout=pd)
    return trunk_cfg


class BiEncoder(torch.nn.Module):
    def __init__(self, config: BiEncoderConfig, device="cuda", seed: Optional[int] = None):
        super().__init__()
        self.config = config
        if not config.encoder:
            raise NotImplementedError("decoder trunks are out of the hot-path scope (SURVEY.md §2a #15)")
        if config.pooling not in ("mean", "cls", "map"):
            # "last" (modeling_biencoder.py:52-77) picks a decoder trunk's eos token; decoder trunks are out of scope
            raise NotImplementedError(f"pooling={config.pooling!r}")
        trunk_cfg = trunk_config_with_overrides(config, config.trunk_config or _default_trunk_config(config.model_name))
        self.is_vision = isinstance(trunk_cfg, ViTConfig)  # image tower: `input_ids` carries the pixel tensor
        if config.pooling == "map" and not self.is_vision:
            raise NotImplementedError("pooling='map' serves the image tower: the reference's masked (text) branch of "
                                      "MultiHeadAttentionPooling does not run (modeling_biencoder.py:134-148)")
        engine_cls = ViTEngine if self.is_vision else NomicBertEngine
        self.trunk = engine_cls(trunk_cfg, device=device, pooling="cls" if config.pooling == "map" else config.pooling,
                                normalize=True, seed=seed)
        # `pooling: map` (configs/train/nomic_embed_vision_v1.5.yaml:69): attention-pooling head above the trunk's hidden states
        self.selector = None
        if config.pooling == "map":
            from .map_pooling import MultiHeadAttentionPooling

            if seed is not None:
                torch.manual_seed(seed + 1)
            self.selector = MultiHeadAttentionPooling(trunk_cfg, device=device)
        if config.gradient_checkpointing:  # modeling_biencoder.py:261-262
            self.trunk.gradient_checkpointing_enable(keep_layers=config.checkpoint_keep_layers)
        self.frozen_trunk = bool(config.freeze)
        self.overlap_reduce = True   # train_args.overlap_grad_reduce: start the gradient all-reduce inside the last backward
        if self.frozen_trunk:
            self.trunk.eval()
            for p in self.trunk.parameters():
                p.requires_grad = False
        d = trunk_cfg.n_embd
        # modeling_biencoder.py:264-267 `proj`: a module compatible with torch's Linear (same `proj.weight` / `proj.bias` keys, same init)
        # whose forward / backward are the HIP bf16 MFMA GEMMs (flash_attn_api FusedDense) -- no vendor BLAS on the product path
        if config.projection_dim:
            from .flash_attn_api.ops.fused_dense import FusedDense

            if int(config.projection_dim) % 4:
                raise NotImplementedError("projection_dim must be a multiple of 4 (row alignment of the GEMM's bf16 output)")
            self.proj = FusedDense(d, int(config.projection_dim), device=device)
        else:
            self.proj = torch.nn.Identity()
        self.hamming = bool(config.hamming)

    @property
    def device(self):
        return self.trunk.device_

    def train(self, mode: bool = True):
        super().train(mode)
        if self.frozen_trunk:
            self.trunk.eval()
        return self

    def forward(self, input_ids, attention_mask=None, is_padded_inputs=True, normalize=True, binarize=False,
                seqlens=None, **kwargs):
        plain = (not self.hamming) and isinstance(self.proj, torch.nn.Identity) and not binarize and self.selector is None
        eng_norm = bool(normalize) and plain
        differentiable = torch.is_grad_enabled() and self.training and not self.frozen_trunk
        if self.selector is not None:  # attention pooling over the trunk's final hidden states
            if differentiable:
                hidden = _VitHiddenFn.apply(self.trunk.flat_decay, self.trunk, input_ids)
            else:
                with torch.no_grad():
                    hidden, _ = self.trunk.forward_hidden_chunk(input_ids, False)
            emb = self.selector(hidden, None, None)
        elif self.is_vision:  # (B, 3, H, W) pixels, no mask (modeling_biencoder.py:84-86)
            if differentiable:
                emb = _VitEncodeFn.apply(self.trunk.flat_decay, self.trunk, input_ids, eng_norm)
            else:
                emb, _ = self.trunk.forward_chunk(input_ids, False, eng_norm)
        else:
            if seqlens is not None:
                vb = VarlenBatch.from_lengths(input_ids, seqlens)
            else:
                vb = VarlenBatch.from_mask(input_ids, attention_mask)
            if differentiable:
                emb = _EncodeFn.apply(self.trunk.flat_decay, self.trunk, vb, eng_norm)
            else:
                emb, _ = self.trunk.forward_chunk(vb, False, eng_norm)
        if not plain:
            if self.hamming:  # LayerNorm without affine on the pooled vector (modeling_biencoder.py:282-285,307)
                if emb.is_cuda:   # the library's LayerNorm kernel (fp32 in / out, fp32 statistics), not an eager torch op
                    from .flash_attn_api.ops.layer_norm import layer_norm as _cx_layer_norm
                    d_ = emb.shape[-1]
                    emb = _cx_layer_norm(emb, torch.ones(d_, device=emb.device), torch.zeros(d_, device=emb.device), 1e-5)
                else:
                    emb = F.layer_norm(emb, (emb.shape[-1],))
            emb = self.proj(emb)
            if binarize:
                emb = (emb > 0).float()
            elif normalize:
                emb = F.normalize(emb, dim=-1)
        return {"embedding": emb, "router_logits": None, "router_loss": None, "tokens_per_expert": None}

    # ---- data-parallel plumbing (what DDP does for the reference, sc/trainers/text_text.py:163-180) -------------
    def arm_overlapped_reduce(self, when_last_outstanding: bool = False):
        """Declare the trunk's next backward (or the one consuming its last saved forward) the step's final gradient
        contribution: its per-block gradient slices are all-reduced on a side stream while the remaining blocks are still
        being differentiated (NomicBertEngine.arm_overlapped_reduce); sync_gradients() then only waits and rescales."""
        if not self.frozen_trunk and self.overlap_reduce:
            self.trunk.arm_overlapped_reduce(when_last_outstanding)

    def sync_gradients(self):
        """Average the flat gradient buffer over ranks (RCCL over xGMI): either the per-block all-reduces an armed final
        backward already put in flight (waited for here), or ONE blocking all-reduce of the whole buffer."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            W = dist.get_world_size()
            marks = getattr(self, "exposed_reduce_marks", None)   # bench.py: a list -> (event, event) per call
            if marks is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()       # completes with the step's last backward kernel (everything queued on the compute stream)
            if not self.trunk.finish_overlapped_reduce():
                dist.all_reduce(self.trunk.flat_grad, op=dist.ReduceOp.SUM)
            self.trunk.flat_grad.div_(W)
            if marks is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()       # the reduced, rescaled gradient exists: e0 -> e1 is the part of the reduction nothing overlapped
                marks.append((e0, e1))
            for p in self._head_parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                    p.grad.div_(W)

    def broadcast_parameters(self, src: int = 0):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.trunk.flat_param, src)
            for p in self._head_parameters():
                dist.broadcast(p.data, src)
            self.trunk.sync_shadows()

    def _head_parameters(self):
        """torch parameters above the flat trunk buffers: the projection and the attention-pooling head."""
        yield from self.proj.parameters()
        if self.selector is not None:
            yield from self.selector.parameters()

    def _head_named_parameters(self):
        yield from (("proj." + n, p) for n, p in self.proj.named_parameters())
        if self.selector is not None:
            yield from (("selector." + n, p) for n, p in self.selector.named_parameters())

    def no_sync(self):  # API parity with DDP-wrapped towers (sc/loss.py:151); reduction is explicit here
        import contextlib

        return contextlib.nullcontext()

    # ---- weights on disk (sc/trainers/base.py:275-290 save_model / load_model; HF `save_pretrained` layout) --------
    def save_pretrained(self, output_dir: str):
        """model.safetensors with the reference's state-dict keys (`trunk.<reference key>`, `proj.*`) + config.json:
        the reference's BiEncoder / eager twin load it unchanged (tests/test_huggingface.py:30-34 key contract)."""
        import dataclasses
        import json
        import os

        from safetensors.torch import save_file

        os.makedirs(output_dir, exist_ok=True)
        sd = {f"trunk.{k}": v.detach().cpu().contiguous() for k, v in self.trunk.reference_state_dict().items()}
        sd.update({f"proj.{k}": v.detach().cpu().contiguous() for k, v in self.proj.state_dict().items()})
        if self.selector is not None:
            sd.update({f"selector.{k}": v.detach().cpu().contiguous() for k, v in self.selector.state_dict().items()})
        save_file(sd, os.path.join(output_dir, "model.safetensors"))
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            json.dump(config_json(self.config, self.trunk.config), f, indent=1)

    def load_pretrained(self, model_path: str, strict: bool = True):
        import os

        from safetensors.torch import load_file

        sd = load_file(os.path.join(model_path, "model.safetensors"))
        self.trunk.load_reference_state_dict({k[6:]: v for k, v in sd.items() if k.startswith("trunk.")}, strict=strict)
        proj = {k[5:]: v for k, v in sd.items() if k.startswith("proj.")}
        if proj:
            self.proj.load_state_dict(proj)
        sel = {k[9:]: v for k, v in sd.items() if k.startswith("selector.")}
        if sel and self.selector is not None:
            self.selector.load_state_dict(sel, strict=strict)
        return self

    def param_groups(self, weight_decay: float):
        """decay / no-decay groups of sc/optimizer.py:16-25 over the flat buffers."""
        groups = [{"params": [self.trunk.flat_decay], "weight_decay": weight_decay},
                  {"params": [self.trunk.flat_nodecay], "weight_decay": 0.0}]
        # sc/optimizer.py:16-25: ndim < 2 after squeeze, "bias" in the name -> no decay (the pooling head's (1, 1, d) latent too)
        proj_w = [p for n, p in self._head_named_parameters() if p.squeeze().ndim >= 2 and "bias" not in n]
        proj_b = [p for n, p in self._head_named_parameters() if not (p.squeeze().ndim >= 2 and "bias" not in n)]
        if proj_w:
            groups[0]["params"] += proj_w
        if proj_b:
            groups[1]["params"] += proj_b
        return groups


class DualEncoder(torch.nn.Module):
    """Two towers + bidirectional gathered InfoNCE (host-side mirror of
    sc/models/dual_encoder/modeling_dual_encoder.py:36-68): both towers are called with normalize=False, L2-normalised,
    all-gathered, and the loss is (CE(v -> all t) + CE(t -> all v)) / 2 * world_size with labels arange(n) + n*rank.
    Each direction is one fused similarity+CE kernel (cx_infonce_fwd/bwd); logits are never materialised.  Towers are
    any modules returning {"embedding": ...}: the native BiEncoder over a NomicBertEngine (text) or a ViTEngine (image)."""

    def __init__(self, text: torch.nn.Module, vision: torch.nn.Module, logit_scale: LogitScale,
                 precomputed_text: bool = False, use_fp8: Optional[bool] = None):
        super().__init__()
        if precomputed_text and not getattr(text, "frozen_trunk", False):
            raise AssertionError("Precomputed text model must be frozen")  # modeling_dual_encoder.py:16-18
        self.text, self.vision, self.logit_scale = text, vision, logit_scale
        self.precomputed_text = precomputed_text
        self.use_fp8 = bool(use_fp8)  # cfg 5 sets `use_fp8: true` (TrainArgs.use_fp8, passed in by ImageTextTrainer)

    def encode_text(self, text, normalize=True):  # modeling_dual_encoder.py:26-29
        return self.text(**text, normalize=normalize)["embedding"]

    def encode_image(self, vision, normalize=True):  # :31-34 (the pixel tensor is passed positionally)
        return self.vision(vision, normalize=normalize)["embedding"]

    def forward(self, text_inputs, vision_inputs):
        from .loss import _infonce, _scale_of

        if self.precomputed_text:  # LiT with text embeddings computed offline (:37-41): the text tower is not run
            if "text_embs" not in text_inputs:
                raise AssertionError("Precomputed text inputs must have text_embs")
            raw_text = text_inputs["text_embs"].to(self.logit_scale.logit_scale.device, torch.float32)
        else:
            raw_text = self.text(**text_inputs, normalize=False)["embedding"]
        text_emb = F.normalize(raw_text, dim=-1, p=2)
        vision_emb = F.normalize(self.vision(**vision_inputs, normalize=False)["embedding"], dim=-1, p=2)
        all_text, all_vis = gather_with_grad(text_emb), gather_with_grad(vision_emb)
        inited = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank() if inited else 0
        world = dist.get_world_size() if inited else 1
        n = vision_emb.shape[0]
        labels = torch.arange(n, device=vision_emb.device) + n * rank
        scale, sp = _scale_of(self.logit_scale)
        coef = 0.5 * world / n
        loss = (_infonce(vision_emb, all_text, labels, scale, coef, sp, self.use_fp8)
                + _infonce(text_emb, all_vis, labels, scale, coef, sp, self.use_fp8))
        return {"loss": loss, "image_text_loss": loss}
