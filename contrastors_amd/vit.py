"""ViT image tower on the native engine (host-side mirror of sc/models/vit/vit.py:127-276 ViTModel,
sc/layers/embedding.py:330-516 PatchEmbedding and the pre-norm branch of sc/layers/block.py:293-388).

`ViTEngine(config)(pixels)` -> (B, d) fp32 pooled (+ L2-normalised) embeddings.  One native call per chunk:
patchify -> patch-projection GEMM -> [cls | patches] + position embeddings -> L pre-norm blocks -> ln_f -> pooling
(cx_vit_forward / cx_vit_backward).  Parameters use the reference's state-dict keys (`embeddings.proj.weight`,
`embeddings.cls_token`, `embeddings.pos_embed`, `layers.{l}.attn.Wqkv.weight`, ..., `ln_f.weight`), so weights remapped
by the reference's `remap_state_dict_hf_vit` load unchanged.

Supported family: the google/vit-* configuration hf_vit_config_to_vit_config produces (sc/models/vit/hf_vit.py:9-53):
pre-norm, GELU MLP with biases, qkv bias, learned absolute position embeddings incl. the cls slot, cls token, no rotary,
dropout / drop-path 0, final LayerNorm.  Anything else raises.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _C
from .nomic_bert import NomicBertEngine, _ChunkArena


@dataclass
class ViTConfig:
    """Fields named like the GPT2Config the reference builds for a ViT (sc/models/vit/hf_vit.py:9-53)."""

    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    n_inner: int = 3072
    activation_function: str = "gelu"
    layer_norm_epsilon: float = 1e-12
    initializer_range: float = 0.02
    img_size: int = 224
    patch_size: int = 16
    num_channels: int = 3
    prenorm: bool = True
    qkv_proj_bias: bool = True
    mlp_fc1_bias: bool = True
    mlp_fc2_bias: bool = True
    patch_embed_bias: bool = True
    # the OpenAI-CLIP image tower (sc/models/vit/clip.py:14-58): activation_function "quick_gelu", a LayerNorm on the
    # embeddings ahead of the first block (`prepre_layernom`, reference spelling), no patch bias, eps 1e-5
    prepre_layernom: bool = False
    rotary_emb_fraction: float = 0.0
    resid_pdrop: float = 0.0
    embd_pdrop: float = 0.0
    attn_pdrop: float = 0.0
    drop_path_rate: float = 0.0
    # PatchDropout (sc/layers/embedding.py:415-418, 519-557): fraction of the patch tokens dropped per image in TRAINING
    # (the [cls] token is always kept); 0 = off
    patch_dropout: float = 0.0

    def __post_init__(self):
        if self.n_embd != self.n_head * 64:
            raise NotImplementedError("head_dim must be 64")
        if self.activation_function not in ("gelu", "gelu_new", "gelu_python", "quick_gelu"):
            raise NotImplementedError(f"activation {self.activation_function!r} (image towers: erf GELU or quick_gelu MLPs)")
        if not self.prenorm:
            raise NotImplementedError("post-norm ViT")
        if any(p != 0 for p in (self.resid_pdrop, self.embd_pdrop, self.attn_pdrop, self.drop_path_rate)):
            raise NotImplementedError("dropout / drop-path > 0")
        if self.rotary_emb_fraction != 0:
            raise NotImplementedError("rotary ViT (eva02)")
        if not 0.0 <= self.patch_dropout < 1.0:
            raise ValueError(f"patch_dropout must be in [0, 1), got {self.patch_dropout}")
        if self.img_size % self.patch_size or self.patch_size % 4:
            raise NotImplementedError("img_size must be a multiple of patch_size, patch_size of 4")
        if self.patch_dim % 64:
            raise NotImplementedError("num_channels * patch_size^2 must be a multiple of 64")

    @property
    def gated(self) -> bool:
        return False

    @property
    def n_patch(self) -> int:
        return (self.img_size // self.patch_size) ** 2

    @property
    def patch_dim(self) -> int:
        return self.num_channels * self.patch_size * self.patch_size

    @classmethod
    def vit_base_patch16_224(cls, **kw) -> "ViTConfig":
        """google/vit-base-patch16-224 (the image tower of BASELINE configs 4 and 5)."""
        return cls(**kw)

    @property
    def mlp_act(self) -> int:   # CxEncoderDesc.mlp_act
        return 1 if self.activation_function == "quick_gelu" else 0

    @classmethod
    def clip_vit_base_patch16(cls, **kw) -> "ViTConfig":
        """openai/clip-vit-base-patch16's image tower through sc/models/vit/clip.py:14-58: quick_gelu, pre-LayerNorm, no
        patch-embedding bias, LayerNorm eps 1e-5."""
        base = dict(activation_function="quick_gelu", prepre_layernom=True, patch_embed_bias=False, layer_norm_epsilon=1e-5)
        base.update(kw)
        return cls(**base)


class ViTEngine(NomicBertEngine):
    """Image trunk + pooling.  Inherits the flat fp32 parameter / gradient buffers, bf16 shadows and chunk arenas."""

    _LAYER_PREFIX = "layers.{l}."

    def __init__(self, config: ViTConfig, device="cuda", pooling: str = "cls", normalize: bool = True,
                 seed: Optional[int] = None):
        super().__init__(config, device=device, pooling=pooling, normalize=normalize, seed=seed)

    # ---- parameter registry ---------------------------------------------------------------------------------------
    def _param_specs(self):
        cfg = self.config
        d = cfg.n_embd
        # decay / no-decay as sc/optimizer.py:16-25 decides them: squeeze().ndim < 2 or "bias" -> no decay
        decay: List[Tuple[str, Tuple[int, ...]]] = [("embeddings.pos_embed", (1, cfg.n_patch + 1, d)),
                                                    ("embeddings.proj.weight", (d, cfg.patch_dim))]
        nodecay: List[Tuple[str, Tuple[int, ...]]] = [("embeddings.cls_token", (1, 1, d))]
        if cfg.patch_embed_bias:
            nodecay.append(("embeddings.proj.bias", (d,)))
        for l in range(cfg.n_layer):
            dl, nl = self._layer_specs(l)
            decay += dl
            nodecay += nl
        nodecay += [("ln_f.weight", (d,)), ("ln_f.bias", (d,))]
        if cfg.prepre_layernom:
            nodecay += [("prepre_layernom.weight", (d,)), ("prepre_layernom.bias", (d,))]
        return decay, nodecay

    def _is_linear(self, name: str) -> bool:
        return name == "embeddings.proj.weight" or super()._is_linear(name)

    def _init_weights(self, seed: Optional[int]):
        """sc/models/vit/vit.py:82-103: Linear normal(0, range) with zero bias, out_proj / fc2 rescaled by
        1/sqrt(2 n_layer); cls_token zeros, pos_embed randn * 0.02 (sc/layers/embedding.py:378-405); LN = (1, 0)."""
        cfg = self.config
        gen = torch.Generator(device="cpu")
        gen.manual_seed(0 if seed is None else seed)
        with torch.no_grad():
            for name, (off, shape) in self._layout.items():
                n = int(np.prod(shape))
                view = self.flat_param[off: off + n]
                if name.endswith(".bias") or name == "embeddings.cls_token":
                    view.zero_()
                elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name in ("ln_f.weight", "prepre_layernom.weight"):
                    view.fill_(1.0)
                else:
                    std = cfg.initializer_range
                    if name.endswith("out_proj.weight") or name.endswith("fc2.weight"):
                        std = cfg.initializer_range / math.sqrt(2 * cfg.n_layer)
                    if name == "embeddings.pos_embed":
                        std = 0.02
                    view.copy_(torch.empty(n, dtype=torch.float32).normal_(0.0, std, generator=gen))

    def _build_rotary(self):
        self.rot_cos = self.rot_sin = None

    def _build_desc(self):
        super()._build_desc()
        e, cfg = self._desc, self.config
        P = lambda n: self.p(n).data_ptr() if n in self._layout else None  # noqa: E731
        G = lambda n: self.g(n).data_ptr() if n in self._layout else None  # noqa: E731
        e.prenorm = 1
        e.lnf_g, e.lnf_b, e.glnf_g, e.glnf_b = P("ln_f.weight"), P("ln_f.bias"), G("ln_f.weight"), G("ln_f.bias")
        e.Wpatch = self._w16("embeddings.proj.weight").data_ptr()
        e.bpatch, e.gbpatch = P("embeddings.proj.bias"), G("embeddings.proj.bias")
        e.cls_token, e.gcls_token = P("embeddings.cls_token"), G("embeddings.cls_token")
        e.vit_pos, e.gvit_pos = P("embeddings.pos_embed"), G("embeddings.pos_embed")
        e.gWpatch = G("embeddings.proj.weight")
        e.patch_dim = cfg.patch_dim
        e.mlp_act = cfg.mlp_act
        e.lnpre_g, e.lnpre_b = P("prepre_layernom.weight"), P("prepre_layernom.bias")
        e.glnpre_g, e.glnpre_b = G("prepre_layernom.weight"), G("prepre_layernom.bias")

    # ---- compute --------------------------------------------------------------------------------------------------
    def _cu_seqlens(self, B: int, S: Optional[int] = None) -> torch.Tensor:
        S = self.config.n_patch + 1 if S is None else S
        key = ("vit_cu", B, S)
        hit = getattr(self, "_cu_cache", {}).get(key)
        if hit is None:
            hit = torch.arange(0, (B + 1) * S, S, dtype=torch.int32).to(self.device_)
            self._cu_cache = getattr(self, "_cu_cache", {})
            self._cu_cache[key] = hit
        return hit

    def _check_pixels(self, pixels: torch.Tensor) -> torch.Tensor:
        cfg = self.config
        if pixels.dim() != 4 or pixels.shape[1] != cfg.num_channels or pixels.shape[2] != cfg.img_size \
                or pixels.shape[3] != cfg.img_size:
            raise ValueError(f"pixels must be (B, {cfg.num_channels}, {cfg.img_size}, {cfg.img_size}); position-embedding "
                             "interpolation for other resolutions is not built")
        if pixels.dtype not in (torch.float32, torch.bfloat16):
            pixels = pixels.float()
        return pixels.contiguous()

    def _patch_subset(self, B: int):
        """PatchDropout (sc/layers/embedding.py:531-557), training only: every image keeps max(1, int(n_patch * (1 - p))) patch
        tokens -- the top-k of a standard-normal draw per patch, taken from torch's CPU generator exactly as the reference does
        (`torch.randn(batch, num_tokens)` then `.topk(k).indices`), in top-k order.  Under GradCache the re-forward replays the
        draw because RandContext restores the CPU generator too.  Returns (keep (B, K) int32, inverse (B, n_patch) int32, K) on
        the device, or None when nothing is dropped."""
        p = float(getattr(self.config, "patch_dropout", 0.0) or 0.0)
        if p <= 0.0 or not self.training:
            return None
        P = self.config.n_patch
        K = max(1, int(P * (1.0 - p)))
        keep = torch.randn(B, P).topk(K, dim=-1).indices
        inv = torch.full((B, P), -1, dtype=torch.int32)
        inv.scatter_(1, keep, torch.arange(K, dtype=torch.int32).expand(B, K))
        return keep.to(torch.int32).to(self.device_), inv.to(self.device_), K

    @property
    def uses_rng(self) -> bool:
        """PatchDropout draws from torch's CPU generator (`_patch_subset`): GradCache's re-forward must replay the draw, so the
        per-chunk RandContext of sc/loss.py:141-145 has to be taken although every dropout probability of the tower is 0
        (without this, pass 2 re-forwarded a different patch subset than the one the cached gradient belongs to)."""
        return self.training and (float(getattr(self.config, "patch_dropout", 0.0) or 0.0) > 0.0 or super().uses_rng)

    def _set_patch_subset(self, arena: _ChunkArena, subset):
        d = arena.desc
        if subset is None:
            d.patch_keep, d.patch_inv, d.n_keep, d.n_patch_all = None, None, 0, 0
            arena.patch_subset = None
        else:
            keep, inv, K = subset
            d.patch_keep, d.patch_inv, d.n_keep, d.n_patch_all = keep.data_ptr(), inv.data_ptr(), K, self.config.n_patch
            arena.patch_subset = subset   # (keeps the index tensors alive until the backward has run)

    def forward_chunk(self, pixels: torch.Tensor, save_for_backward: bool, normalize: Optional[bool] = None,
                      out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[_ChunkArena]]:
        cfg = self.config
        pixels = self._check_pixels(pixels)
        B = pixels.shape[0]
        subset = self._patch_subset(B)
        S = (subset[2] if subset else cfg.n_patch) + 1
        T = B * S
        if out is None:
            out = torch.empty(B, cfg.n_embd, dtype=torch.float32, device=self.device_)
        arena = self._get_arena(T, B, save_for_backward)
        self._set_patch_subset(arena, subset)
        self._desc.normalize = int(self.normalize_default if normalize is None else normalize)
        rc = self.lib.cx_vit_forward(C.byref(self._desc), C.byref(arena.desc), pixels.data_ptr(),
                                     int(pixels.dtype == torch.bfloat16), self._cu_seqlens(B, S).data_ptr(), B,
                                     cfg.num_channels, cfg.img_size, cfg.img_size, cfg.patch_size,
                                     int(save_for_backward), out.data_ptr(), _C.cur_stream())
        _C.check(rc, "cx_vit_forward")
        if save_for_backward:
            arena.emb_out = out
            arena.normalize = self._desc.normalize
            self._outstanding += 1
            return out, arena
        return out, None

    def backward_chunk(self, pixels_or_B, arena: _ChunkArena, demb: torch.Tensor):
        assert arena.emb_out is not None, "backward_chunk needs a forward with save_for_backward=True"
        B = pixels_or_B if isinstance(pixels_or_B, int) else pixels_or_B.shape[0]
        demb = demb.to(torch.float32).contiguous()
        self._desc.normalize = arena.normalize
        fires = self._begin_backward(arena)
        sub = getattr(arena, "patch_subset", None)
        P = sub[2] if sub else self.config.n_patch
        rc = self.lib.cx_vit_backward(C.byref(self._desc), C.byref(arena.desc), self._cu_seqlens(B, P + 1).data_ptr(), B,
                                      P, demb.data_ptr(), arena.emb_out.data_ptr(), _C.cur_stream())
        _C.check(rc, "cx_vit_backward")
        arena.patch_subset = None
        self._end_backward(arena, fires)

    # ---- token-level outputs (poolers above the C-ABI: `pooling: map`): (B, n_patch + 1, d) bf16 after ln_f ------------
    def forward_hidden_chunk(self, pixels: torch.Tensor, save_for_backward: bool):
        cfg = self.config
        pixels = self._check_pixels(pixels)
        B = pixels.shape[0]
        subset = self._patch_subset(B)
        S = (subset[2] if subset else cfg.n_patch) + 1
        hidden = torch.empty(B, S, cfg.n_embd, dtype=torch.bfloat16, device=self.device_)
        arena = self._get_arena(B * S, B, save_for_backward)
        self._set_patch_subset(arena, subset)
        rc = self.lib.cx_vit_forward_hidden(C.byref(self._desc), C.byref(arena.desc), pixels.data_ptr(),
                                            int(pixels.dtype == torch.bfloat16), self._cu_seqlens(B, S).data_ptr(), B,
                                            cfg.num_channels, cfg.img_size, cfg.img_size, cfg.patch_size,
                                            int(save_for_backward), hidden.data_ptr(), _C.cur_stream())
        _C.check(rc, "cx_vit_forward_hidden")
        if save_for_backward:
            arena.emb_out = hidden   # (marks the arena as holding a saved forward)
            self._outstanding += 1
            return hidden, arena
        return hidden, None

    def backward_hidden_chunk(self, B: int, arena: _ChunkArena, dhidden: torch.Tensor):
        dh = dhidden.to(torch.bfloat16).contiguous()
        fires = self._begin_backward(arena)
        sub = getattr(arena, "patch_subset", None)
        P = sub[2] if sub else self.config.n_patch
        rc = self.lib.cx_vit_backward_hidden(C.byref(self._desc), C.byref(arena.desc), self._cu_seqlens(B, P + 1).data_ptr(), B,
                                             P, dh.data_ptr(), _C.cur_stream())
        _C.check(rc, "cx_vit_backward_hidden")
        arena.patch_subset = None
        self._end_backward(arena, fires)

    def hidden_states(self, pixels: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training:
            return _VitHiddenFn.apply(self.flat_decay, self, pixels)
        return self.forward_hidden_chunk(pixels, False)[0]

    def forward(self, pixels: torch.Tensor, attention_mask=None, normalize: Optional[bool] = None) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training:
            return _VitEncodeFn.apply(self.flat_decay, self, pixels, normalize)
        emb, _ = self.forward_chunk(pixels, False, normalize)
        return emb


class _VitEncodeFn(torch.autograd.Function):
    """autograd bridge (see nomic_bert._EncodeFn): backward accumulates into the engine's flat gradient buffer."""

    @staticmethod
    def forward(ctx, _anchor, engine: ViTEngine, pixels: torch.Tensor, normalize):
        emb, arena = engine.forward_chunk(pixels, True, normalize)
        ctx.engine, ctx.B, ctx.arena = engine, pixels.shape[0], arena
        return emb

    @staticmethod
    def backward(ctx, demb):
        arena, ctx.arena = ctx.arena, None   # (a graph that outlives its backward must not keep the arena alive)
        ctx.engine.backward_chunk(ctx.B, arena, demb)
        return None, None, None, None


class _VitHiddenFn(torch.autograd.Function):
    """Token-level twin of _VitEncodeFn: hidden states out, d(hidden) in, parameter gradients into the flat buffer."""

    @staticmethod
    def forward(ctx, _anchor, engine: ViTEngine, pixels: torch.Tensor):
        hidden, arena = engine.forward_hidden_chunk(pixels, True)
        ctx.engine, ctx.B, ctx.arena = engine, pixels.shape[0], arena
        return hidden

    @staticmethod
    def backward(ctx, dhidden):
        arena, ctx.arena = ctx.arena, None
        ctx.engine.backward_hidden_chunk(ctx.B, arena, dhidden)
        return None, None, None
