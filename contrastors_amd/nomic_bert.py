"""NomicBert / BERT encoder on the native MI355X engine.

Host-side mirror of the reference's flash model (sc/models/encoder/modeling_nomic_bert.py:488-587 `NomicBertModel`,
configuration_nomic_bert.py, and the Block/FlashAttention/MLP python it drives): same config field names, same
state-dict keys, but the whole forward / backward of a chunk is ONE call into
libcontrastors_hip.so (cx_encoder_forward / cx_encoder_backward) over a caller-owned activation arena.

Memory layout (sized for 288 GB HBM3E):
  * fp32 master parameters live in ONE flat buffer `[decay segment | no-decay segment]` (optimizer.py:16-25 grouping),
    gradients in a second flat buffer of the same layout -> the optimizer, grad-clip and the DP all-reduce each touch
    two big tensors instead of ~150 small ones;
  * bf16 shadows of every Linear weight (row-major for fwd, transposed for dgrad) are refreshed once per optimizer
    step (`sync_shadows`), replacing the reference's per-chunk autocast re-cast;
  * activations of a chunk live in per-layer slots of a pre-allocated arena (`_ChunkArena`).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import logging
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.distributed as dist

from . import _C
from .policy import _parse_keep


@dataclass
class NomicBertConfig:
    """Field names follow sc/models/encoder/configuration_nomic_bert.py:4-56 (GPT2Config naming)."""

    vocab_size: int = 30528
    n_positions: int = 2048
    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    n_inner: int = 3072
    activation_function: str = "swiglu"
    rotary_emb_fraction: float = 1.0
    rotary_emb_base: float = 1000.0
    rotary_emb_interleaved: bool = False
    qkv_proj_bias: bool = False
    mlp_fc1_bias: bool = False
    mlp_fc2_bias: bool = False
    prenorm: bool = False
    layer_norm_epsilon: float = 1e-12
    type_vocab_size: int = 2
    pad_token_id: int = 0
    initializer_range: float = 0.02
    resid_pdrop: float = 0.0
    embd_pdrop: float = 0.0
    attn_pdrop: float = 0.0
    causal: bool = False
    use_rms_norm: bool = False
    max_position_embeddings: int = 512  # learned absolute positions when rotary_emb_fraction == 0
    rotary_scaling_factor: Optional[float] = None  # Dynamic-NTK rotary beyond max_trained_positions (attention.py:52-60)
    max_trained_positions: int = 2048

    def __post_init__(self):
        if self.prenorm or self.causal or self.use_rms_norm or self.rotary_emb_interleaved:
            raise NotImplementedError("engine covers the post-norm, non-causal, LayerNorm encoder (cfg 1-3)")
        if not (0.0 <= self.resid_pdrop < 1.0 and 0.0 <= self.embd_pdrop < 1.0 and 0.0 <= self.attn_pdrop < 1.0):
            raise ValueError("dropout probabilities must be in [0, 1)")
        if self.n_embd != self.n_head * 64:
            raise NotImplementedError("head_dim must be 64")
        if self.rotary_emb_fraction not in (0.0, 1.0):
            raise NotImplementedError("rotary_emb_fraction must be 0 or 1")
        if self.activation_function not in ("swiglu", "gelu", "gelu_new"):
            raise NotImplementedError(self.activation_function)
        if self.n_inner % 32:
            raise NotImplementedError("n_inner must be a multiple of 32 (interleaved fc11/fc12 layout)")

    @property
    def gated(self) -> bool:
        return self.activation_function == "swiglu"

    @property
    def hidden_size(self) -> int:
        return self.n_embd

    @classmethod
    def nomic_bert_2048(cls, **kw) -> "NomicBertConfig":
        """cfg 2/3 constants: SURVEY.md §8(d) / Appendix D (mlm.yaml:33-46, bert.py:11-50)."""
        return cls(**kw)

    @classmethod
    def bert_base_uncased(cls, hf_dropout: bool = False, **kw) -> "NomicBertConfig":
        """cfg 1: bert_config_to_nomic_config of the standard HF BertConfig (sc/models/encoder/bert.py:11-50).
        `hf_dropout`: that conversion carries BertConfig's hidden_dropout_prob / attention_probs_dropout_prob = 0.1 into
        resid_pdrop / embd_pdrop / attn_pdrop (bert.py:19-21), so a reference run on bert-base-uncased TRAINS WITH DROPOUT
        0.1; True reproduces that (what a recipe naming the model gets, biencoder._default_trunk_config), False is the
        architecture alone (parity tests against the oracle, throughput legs; pinned to the reference function in
        tests/test_host_cpu.py)."""
        base = dict(vocab_size=30522, n_positions=512, max_position_embeddings=512, activation_function="gelu",
                    rotary_emb_fraction=0.0, qkv_proj_bias=True, mlp_fc1_bias=True, mlp_fc2_bias=True)
        if hf_dropout:
            base.update(resid_pdrop=0.1, embd_pdrop=0.1, attn_pdrop=0.1)
        base.update(kw)
        return cls(**base)


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def parse_checkpoint_keep(v, where: str, default: Union[int, str] = 0) -> Union[int, str]:
    """checkpoint_keep_layers: a non-negative integer or "auto" (None / "" -> `default`); see policy._parse_keep."""
    return default if v is None or v == "" else _parse_keep(v, where)


_HBM_GRANTED: Dict[int, int] = {}


def _hbm_grant(device, delta: int) -> int:
    """Per-device ledger of the bytes selective checkpointing has promised to (or already built into) arenas beyond their
    literal-checkpointing size.  Returns the balance after adding `delta`."""
    idx = device.index if getattr(device, "index", None) is not None else 0
    _HBM_GRANTED[idx] = max(0, _HBM_GRANTED.get(idx, 0) + int(delta))
    return _HBM_GRANTED[idx]


class _ChunkArena:
    """Device memory for one in-flight chunk (CxChunkBuffers).  `n_slots` = 1 for no-grad forwards, n_layer else."""

    probation = False   # "auto" selective checkpointing: first use, literal; its successor is sized from the measured peak
    last_tick = 0       # the engine's saving-forward counter when this arena was last handed out (idle arenas are given back)
    granted = 0         # bytes of the per-device ledger this arena holds (returned when it is destroyed)

    def __del__(self):
        if self.granted:
            try:
                _hbm_grant(self._device, -self.granted)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass

    def __init__(self, cfg: NomicBertConfig, T_cap: int, n_slots: int, with_backward: bool, B_cap: int,
                 device: torch.device, checkpoint: bool = False, keep_layers: int = 0):
        d, I, H = cfg.n_embd, cfg.n_inner, cfg.n_head
        wfc1 = 2 * I if cfg.gated else I
        bf = dict(dtype=torch.bfloat16, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        self.T_cap, self.n_slots, self.with_backward, self.B_cap = T_cap, n_slots, with_backward, B_cap
        self._device = device
        self.checkpoint = bool(checkpoint and with_backward)
        # activation checkpointing: only the tensor carrying a block's input keeps one slot per layer (CxChunkBuffers);
        # selective checkpointing: the top `keep_layers` blocks keep everything (slots 1 .. keep), the rest share slot 0
        self.keep_layers = max(0, min(int(keep_layers), n_slots)) if self.checkpoint else 0
        kept = ("z1" if getattr(cfg, "prenorm", False) else "h2") if self.checkpoint else None
        n_layer_slots = n_slots
        if self.checkpoint:
            n_slots = 1 + self.keep_layers
        t: Dict[str, torch.Tensor] = {}
        t["h0"] = torch.empty(T_cap, d, **bf)
        t["emb_mean"] = torch.empty(T_cap, **f32)
        t["emb_rstd"] = torch.empty(T_cap, **f32)
        t["qkv"] = torch.empty(n_slots, T_cap, 3 * d, **bf)
        t["ctx"] = torch.empty(n_slots, T_cap, d, **bf)
        t["lse"] = torch.empty(n_slots, T_cap * H, **f32)
        for n in ("z1", "h1", "z2", "h2"):
            t[n] = torch.empty(n_layer_slots if n == kept else n_slots, T_cap, d, **bf)
        for n in ("mean1", "rstd1", "mean2", "rstd2"):
            t[n] = torch.empty(n_slots, T_cap, **f32)
        t["yg"] = torch.empty(n_slots, T_cap, I, **bf)   # plain MLP: biased pre-activation; gated MLP: the gate alone (ABI 6)
        t["act"] = torch.empty(n_slots, T_cap, I, **bf)
        t["pool_norm"] = torch.empty(B_cap, **f32)
        patch_dim = int(getattr(cfg, "patch_dim", 0) or 0)
        if getattr(cfg, "prenorm", False):  # input / output / statistics of the final LayerNorm
            t["zf"] = torch.empty(T_cap, d, **bf)
            t["hf"] = torch.empty(T_cap, d, **bf)
            t["meanf"] = torch.empty(T_cap, **f32)
            t["rstdf"] = torch.empty(T_cap, **f32)
        if patch_dim:  # ViT front end: patchified pixels and their projection
            t["patch_in"] = torch.empty(T_cap, patch_dim, **bf)
            t["patch_proj"] = torch.empty(T_cap, d, **bf)
        if getattr(cfg, "prepre_layernom", False):  # CLIP flavour: the pre-LayerNorm's input, kept for its backward
            t["zpre"] = torch.empty(T_cap, d, **bf)
        # split-K workspace (fp32 partial slabs): 8 slabs of the largest weight, 16 of the smallest, for the wgrad GEMMs; every
        # arena has it, because the few-tile long-K projections of SMALL chunks take a split-K route too
        # (cx_gemm_bf16_nt_splitk) and a no-grad forward must round exactly like the saving forward of the same chunk
        # ... and, for a plain (GELU) MLP with a backward, the fc1 bias gradient's per-128-row partials of the fused fc2-dgrad + activation
        # backward (cx_gemm_bf16_act_bwd, round 6): ceil(T / 128) x I floats -- at the CLIP leg's image tower (807 k token rows) 3 % more than
        # the split-K slabs; without it that launch fell back to the two-kernel route
        ws_floats = max(8 * max(3 * d, wfc1, patch_dim) * d, 16 * d * d)
        if with_backward and wfc1 == I:
            ws_floats = max(ws_floats, ((T_cap + 127) // 128) * I)
        t["ws_f32"] = torch.empty(ws_floats, **f32)
        if with_backward:
            wide = max(3 * d, wfc1, patch_dim)
            for n in ("g_a", "g_b", "g_c"):
                t[n] = torch.empty(T_cap, d, **bf)
            t["g_wide"] = torch.empty(T_cap, wide, **bf)
            t["g_act"] = torch.empty(T_cap, I, **bf)
            # transposed operands of the wgrad FALLBACK: the natural-layout kernel covers feature counts that tile by 256 (every
            # BASELINE tower); only other widths (the tiny test trunks) transpose -- 2 x wide x 2 B per token not allocated
            if any(f % 256 for f in (d, 3 * d, I, wfc1) + ((patch_dim,) if patch_dim else ())):
                t["tr_a"] = torch.empty(wide, T_cap, **bf)
                t["tr_b"] = torch.empty(wide, T_cap, **bf)
            t["delta"] = torch.empty(T_cap * H, **f32)
            if getattr(cfg, "resid_pdrop", 0.0) > 0:  # dropout: the LayerNorm backward returns a second gradient
                t["g_d"] = torch.empty(T_cap, d, **bf)
        self.tensors = t
        self.desc = _C.CxChunkBuffers()
        self.desc.T_cap = T_cap
        for name, _ in _C.CxChunkBuffers._fields_[1:]:
            if name == "ws_floats":
                self.desc.ws_floats = t["ws_f32"].numel() if "ws_f32" in t else 0
            elif name == "checkpoint":
                self.desc.checkpoint = int(self.checkpoint)
            elif name == "ckpt_keep":
                self.desc.ckpt_keep = int(self.keep_layers)
            elif name in ("drop_active", "drop_seed", "drop_offset", "n_keep", "n_patch_all"):
                setattr(self.desc, name, 0)
            else:
                setattr(self.desc, name, t[name].data_ptr() if name in t else None)
        self.emb_out: Optional[torch.Tensor] = None  # set by a saving forward, consumed by backward

    def nbytes(self) -> int:
        return sum(x.numel() * x.element_size() for x in self.tensors.values())

    @staticmethod
    def slot_bytes_per_token(cfg) -> int:
        """What ONE more slot of the per-layer buffers costs per token (the price of keeping a block under selective
        checkpointing): qkv, ctx, lse, three of z1 / h1 / z2 / h2 (the fourth is the per-layer input tensor a checkpointing
        arena holds anyway), the LayerNorm statistics, the fc1 pre-activation and the activation."""
        d, I, H = cfg.n_embd, cfg.n_inner, cfg.n_head
        return 2 * (3 * d + d + 3 * d + I + I) + 4 * (H + 4)


_FULL_CACHE: Dict[tuple, tuple] = {}


@dataclass
class VarlenBatch:
    """Token-stream view of a right- or arbitrarily-padded (B,S) batch (flash_attn.bert_padding.unpad_input)."""

    input_ids: torch.Tensor   # (B,S) int64, device
    indices: torch.Tensor     # (T,) int32, device: flat positions b*S+s of the kept tokens
    cu_seqlens: torch.Tensor  # (B+1,) int32, device
    B: int
    S: int
    T: int
    max_seqlen: int
    _sort: Optional[tuple] = None  # (sorted ids int32, permutation int32): built on first use by a backward

    def embedding_sort(self):
        """Token ids of the chunk in ascending stable order + the permutation that sorts them: lets the engine reduce the
        word-embedding gradient per vocabulary row without atomics (cx_embed_ln_bwd_sorted).  One device sort, no sync."""
        if self._sort is None:
            import os

            if os.environ.get("CX_EMBED_ATOMICS"):  # debugging aid: fall back to the fp32-atomics scatter
                self._sort = (None, None)
                return self._sort
            tok = self.input_ids.reshape(-1)[self.indices.long()].to(torch.int32)
            sorted_ids, perm = torch.sort(tok, stable=True)
            self._sort = (sorted_ids.contiguous(), perm.to(torch.int32).contiguous())
        return self._sort

    @staticmethod
    def from_lengths(input_ids: torch.Tensor, seqlens) -> "VarlenBatch":
        """Right-padded batch with host-known lengths: no device->host sync (the dataloader knows the lengths)."""
        B, S = input_ids.shape
        lens = np.asarray(seqlens, dtype=np.int64).reshape(B)
        dev = input_ids.device
        full = bool((lens == S).all())
        if full:  # throughput case: every chunk shares the same index tensors, build them once per (B,S,device)
            hit = _FULL_CACHE.get((B, S, str(dev)))
            if hit is not None:
                return VarlenBatch(input_ids, hit[0], hit[1], B, S, B * S, S)
        cu = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(lens, out=cu[1:])
        T = int(cu[-1])
        if full:
            idx = np.arange(B * S, dtype=np.int32)
        else:
            idx = np.concatenate([b * S + np.arange(l, dtype=np.int32) for b, l in enumerate(lens)]).astype(np.int32)
        idx_t = torch.from_numpy(idx).to(dev, non_blocking=True)
        cu_t = torch.from_numpy(cu).to(dev, non_blocking=True)
        if full:
            _FULL_CACHE[(B, S, str(dev))] = (idx_t, cu_t)
        return VarlenBatch(input_ids, idx_t, cu_t, B, S, T, int(lens.max()) if B else 0)

    @staticmethod
    def from_mask(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> "VarlenBatch":
        """Arbitrary mask, semantics of unpad_input.  Costs one host sync like the reference."""
        B, S = input_ids.shape
        if attention_mask is None:
            return VarlenBatch.from_lengths(input_ids, [S] * B)
        mask = attention_mask.to(torch.bool)
        lens = mask.sum(-1, dtype=torch.int32)
        idx = torch.nonzero(mask.flatten(), as_tuple=False).flatten().to(torch.int32)
        cu = torch.zeros(B + 1, dtype=torch.int32, device=input_ids.device)
        cu[1:] = torch.cumsum(lens, 0)
        return VarlenBatch(input_ids, idx, cu, B, S, int(idx.numel()), int(lens.max().item()) if B else 0)


class NomicBertEngine(torch.nn.Module):
    """Encoder trunk + pooling on libcontrastors_hip.so.  Parameters are views into flat fp32 buffers."""

    def __init__(self, config: NomicBertConfig, device="cuda", pooling: str = "mean", normalize: bool = True,
                 seed: Optional[int] = None):
        super().__init__()
        self.config = config
        self.device_ = torch.device(device)
        if self.device_.type != "cuda":
            raise RuntimeError("NomicBertEngine needs an MI355X (cuda/hip device); there is no CPU path")
        self.lib = _C.lib()
        if pooling not in ("mean", "cls"):
            raise NotImplementedError(f"pooling={pooling}")
        self.pool_mode = 0 if pooling == "mean" else 1
        self.normalize_default = bool(normalize)
        decay, nodecay = self._param_specs()
        self._layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shape in decay:
            self._layout[name] = (off, shape)
            off += _round_up(int(np.prod(shape)), 64)
        self.n_decay = off
        for name, shape in nodecay:
            self._layout[name] = (off, shape)
            off += _round_up(int(np.prod(shape)), 64)
        self.n_total = off
        self._linear_names = [n for n, s in decay if self._is_linear(n)]
        self._lin_begin = self._layout[self._linear_names[0]][0]
        self._lin_end = self.n_decay

        dev = self.device_
        self.flat_param = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        # what torch.optim sees: two parameters (decay / no-decay) whose .grad alias the flat gradient buffer
        self.flat_decay = torch.nn.Parameter(self.flat_param[: self.n_decay])
        self.flat_nodecay = torch.nn.Parameter(self.flat_param[self.n_decay:])
        self.flat_decay.grad = self.flat_grad[: self.n_decay]
        self.flat_nodecay.grad = self.flat_grad[self.n_decay:]
        # bf16 shadows of the Linear weights (same relative layout) + transposed copies
        n_lin = self._lin_end - self._lin_begin
        self.w16 = torch.zeros(n_lin, dtype=torch.bfloat16, device=dev)
        self.w16_t = torch.zeros(n_lin, dtype=torch.bfloat16, device=dev)

        self._init_weights(seed)
        self._build_rotary()
        self._build_desc()
        self._arena_nograd: Optional[_ChunkArena] = None
        self._arena_free: List[_ChunkArena] = []
        self._outstanding = 0          # saving forwards whose backward has not run yet
        self._overlap_armed = None     # see arm_overlapped_reduce
        self._ov_works = None
        # BiEncoderConfig.gradient_checkpointing (sc/models/biencoder/modeling_biencoder.py:261-262): a saving forward
        # keeps one (T, d) tensor per block and backward recomputes the rest block by block (engine.hip, slot mode 2)
        self.gradient_checkpointing = False
        self.checkpoint_keep: Union[int, str] = 0   # blocks that keep their activations under checkpointing: n | "auto"
        self._keep_logged: set = set()
        self._keep_suspended = 0   # > 0: "auto" keeps nothing (a caller that lines up MANY arenas budgets the HBM itself)
        self._keep_plan: Dict[int, int] = {}      # T_cap -> blocks the next arena of that size keeps ("auto", measured)
        self._keep_granted: Dict[int, int] = {}   # T_cap -> ledger bytes promised to that arena, until it is built
        self._keep_granted_B: Dict[int, int] = {}  # T_cap -> sequence capacity of the measured arena (its successor keeps it)
        self._arena_tick = 0                      # saving forwards so far (ages the idle arenas)
        self.sync_shadows()

    # ------------------------------------------------------------------------------------------------ parameters
    _LAYER_PREFIX = "encoder.layers.{l}."

    def _layer_specs(self, l: int):
        """(decay, no-decay) parameter specs of transformer block l, keyed like the reference."""
        cfg = self.config
        d, I = cfg.n_embd, cfg.n_inner
        wfc1 = 2 * I if cfg.gated else I
        p = self._LAYER_PREFIX.format(l=l)
        decay = [(p + "attn.Wqkv.weight", (3 * d, d)), (p + "attn.out_proj.weight", (d, d)),
                 (p + "mlp.fc1_fused.weight", (wfc1, d)), (p + "mlp.fc2.weight", (d, I))]
        nodecay = []
        if cfg.qkv_proj_bias:
            nodecay += [(p + "attn.Wqkv.bias", (3 * d,)), (p + "attn.out_proj.bias", (d,))]
        if cfg.mlp_fc1_bias:
            nodecay.append((p + "mlp.fc1_fused.bias", (wfc1,)))
        if cfg.mlp_fc2_bias:
            nodecay.append((p + "mlp.fc2.bias", (d,)))
        nodecay += [(p + "norm1.weight", (d,)), (p + "norm1.bias", (d,)),
                    (p + "norm2.weight", (d,)), (p + "norm2.bias", (d,))]
        return decay, nodecay

    def _param_specs(self):
        """Parameter registry: (storage name, shape) lists for the decay / no-decay groups of sc/optimizer.py:16-25.
        Every Linear weight that needs a bf16 shadow sits at the END of the decay list (one contiguous region)."""
        cfg = self.config
        d = cfg.n_embd
        decay: List[Tuple[str, Tuple[int, ...]]] = [("embeddings.word_embeddings.weight", (cfg.vocab_size, d))]
        if cfg.rotary_emb_fraction == 0.0:
            decay.append(("embeddings.position_embeddings.weight", (cfg.max_position_embeddings, d)))
        decay.append(("embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, d)))
        nodecay: List[Tuple[str, Tuple[int, ...]]] = [("emb_ln.weight", (d,)), ("emb_ln.bias", (d,))]
        for l in range(cfg.n_layer):
            dl, nl = self._layer_specs(l)
            decay += dl
            nodecay += nl
        return decay, nodecay

    def _is_linear(self, name: str) -> bool:
        return ".layers." in name or name.startswith("layers.")

    def p(self, name: str) -> torch.Tensor:
        off, shape = self._layout[name]
        return self.flat_param[off: off + int(np.prod(shape))].view(shape)

    def g(self, name: str) -> torch.Tensor:
        off, shape = self._layout[name]
        return self.flat_grad[off: off + int(np.prod(shape))].view(shape)

    def _w16(self, name: str, transposed: bool = False) -> torch.Tensor:
        off, shape = self._layout[name]
        buf = self.w16_t if transposed else self.w16
        o = off - self._lin_begin
        shp = (shape[1], shape[0]) if transposed else shape
        return buf[o: o + int(np.prod(shape))].view(shp)

    def _init_weights(self, seed: Optional[int]):
        """normal(0, initializer_range) for Linear/Embedding weights, zero biases, LN = (1, 0), padding row zero
        (sc/models/encoder/modeling_nomic_bert.py:284-292)."""
        cfg = self.config
        gen = torch.Generator(device="cpu")
        gen.manual_seed(0 if seed is None else seed)
        with torch.no_grad():
            for name, (off, shape) in self._layout.items():
                n = int(np.prod(shape))
                view = self.flat_param[off: off + n]
                if name.endswith(".bias"):
                    view.zero_()
                elif len(shape) == 1:
                    view.fill_(1.0)
                else:
                    w = torch.empty(shape, dtype=torch.float32).normal_(0.0, cfg.initializer_range, generator=gen)
                    if name == "embeddings.word_embeddings.weight":
                        w[cfg.pad_token_id].zero_()
                    view.copy_(w.flatten())

    # The gated MLP stores fc11/fc12 as ONE (2I, d) matrix whose rows are interleaved in groups of 32
    # ([fc11 0..31 | fc12 0..31 | fc11 32..63 | ...]) so that the fused GEMM epilogue finds y and gate of the same
    # activation column in the same lane (cx_gemm_bf16_swiglu).  These helpers translate to the reference's keys.
    def _split_fused(self, t: torch.Tensor):
        I = self.config.n_inner
        v = t.view(I // 32, 2, 32, *t.shape[1:])
        return v[:, 0].reshape(I, *t.shape[1:]), v[:, 1].reshape(I, *t.shape[1:])

    def _export(self, getter) -> Dict[str, torch.Tensor]:
        out: Dict[str, torch.Tensor] = {}
        for name in self._layout:
            t = getter(name)
            if ".mlp.fc1_fused." in name:
                if self.config.gated:
                    y, g = self._split_fused(t)
                    out[name.replace("fc1_fused", "fc11")] = y
                    out[name.replace("fc1_fused", "fc12")] = g
                else:
                    out[name.replace("fc1_fused", "fc1")] = t
            else:
                out[name] = t
        return out

    def reference_state_dict(self) -> Dict[str, torch.Tensor]:
        """Parameters keyed like the reference (Appendix E).  fc11/fc12 entries are copies (de-interleaved)."""
        return self._export(self.p)

    def reference_grad_dict(self) -> Dict[str, torch.Tensor]:
        return self._export(self.g)

    @torch.no_grad()
    def load_reference_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Copy weights keyed like the reference flash model / its eager twin (tests/test_huggingface.py:30-34)."""
        I = self.config.n_inner
        missing = []
        for name in self._layout:
            dst = self.p(name)
            if ".mlp.fc1_fused." in name and self.config.gated:
                k1, k2 = name.replace("fc1_fused", "fc11"), name.replace("fc1_fused", "fc12")
                if k1 not in sd or k2 not in sd:
                    missing.append(k1)
                    continue
                v = dst.view(I // 32, 2, 32, *dst.shape[1:])
                v[:, 0].copy_(sd[k1].to(device=dst.device, dtype=torch.float32).view(I // 32, 32, *dst.shape[1:]))
                v[:, 1].copy_(sd[k2].to(device=dst.device, dtype=torch.float32).view(I // 32, 32, *dst.shape[1:]))
                continue
            key = name.replace("fc1_fused", "fc1")
            if key not in sd:
                missing.append(key)
                continue
            src = sd[key]
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"{key}: shape {tuple(src.shape)} != {tuple(dst.shape)}")
            dst.copy_(src.to(device=dst.device, dtype=torch.float32))
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}...")
        self.sync_shadows()

    @torch.no_grad()
    def sync_shadows(self):
        """Refresh the bf16 (and transposed bf16) copies of every Linear weight from the fp32 masters."""
        s = _C.cur_stream()
        n = self._lin_end - self._lin_begin
        _C.check(self.lib.cx_cast_f32_to_bf16(self.flat_param.data_ptr() + 4 * self._lin_begin, self.w16.data_ptr(),
                                               n, s), "cast")
        jobs = getattr(self, "_cast_jobs", None)
        if jobs is not None and (jobs[3] != self.flat_param.data_ptr() or jobs[4] != self.w16_t.data_ptr()):
            jobs = None    #  the buffers were re-bound (.to(), a load that replaced them): the table holds raw pointers
        if jobs is None:   # one launch for every transposed shadow (CxCastJob table in device memory)
            tab = np.zeros(len(self._linear_names), dtype=np.dtype([("in", "u8"), ("out", "u8"), ("rows", "i4"), ("cols", "i4")]))
            tiles = 0
            for i, name in enumerate(self._linear_names):
                off, shape = self._layout[name]
                tab[i] = (self.flat_param.data_ptr() + 4 * off, self._w16(name, True).data_ptr(), shape[0], shape[1])
                tiles = max(tiles, ((shape[0] + 63) // 64) * ((shape[1] + 63) // 64))
            dev_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device_)
            jobs = self._cast_jobs = (dev_tab, len(self._linear_names), tiles, self.flat_param.data_ptr(), self.w16_t.data_ptr())
        _C.check(self.lib.cx_cast_transpose_f32_to_bf16_batched(jobs[0].data_ptr(), jobs[1], jobs[2], s), "cast_transpose")

    def zero_grad(self, set_to_none: bool = False):  # noqa: D401 - torch signature
        self.flat_grad.zero_()

    # ------------------------------------------------------------------------------------------------ descriptors
    def _build_rotary(self):
        cfg = self.config
        self.rot_cos = self.rot_sin = None
        self._rot_len = 0
        if cfg.rotary_emb_fraction > 0:
            # flash_attn RotaryEmbedding: inv_freq = 1/base^(arange(0,dim,2)/dim), fp32
            self._inv_freq = self._rot_inv_freq(cfg.rotary_emb_base)
            # Dynamic NTK builds its table on demand (the reference's cache does too); plain rotary covers n_positions
            self._rot_tables(0 if cfg.rotary_scaling_factor else cfg.n_positions)

    @staticmethod
    def _rot_inv_freq(base: float) -> torch.Tensor:
        dim = 64
        return 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))

    def _rot_tables(self, seqlen: int):
        if seqlen <= 0:
            return
        freqs = torch.outer(torch.arange(seqlen, dtype=torch.float32), self._inv_freq)
        self.rot_cos = torch.cos(freqs).contiguous().to(self.device_)
        self.rot_sin = torch.sin(freqs).contiguous().to(self.device_)
        self._rot_len = seqlen
        if getattr(self, "_desc", None) is not None:
            self._desc.rot_cos, self._desc.rot_sin = self.rot_cos.data_ptr(), self.rot_sin.data_ptr()
            self._desc.max_pos = seqlen

    def _update_rotary(self, max_seqlen: int):
        """DynamicNTKRotaryEmbedding._update_cos_sin_cache (sc/layers/embedding.py:810-865) with its state machine: a
        batch longer than max_trained_positions re-bases inv_freq (and that stays), the table is rebuilt only when a
        longer sequence than the cached one arrives."""
        cfg = self.config
        if not cfg.rotary_scaling_factor or self.rot_cos is None and cfg.rotary_emb_fraction == 0:
            return
        f, mx, dim = cfg.rotary_scaling_factor, cfg.max_trained_positions, 64
        if max_seqlen > mx:
            self._inv_freq = self._rot_inv_freq(cfg.rotary_emb_base * ((f * max_seqlen / mx) - (f - 1)) ** (dim / (dim - 2)))
        if max_seqlen > self._rot_len:
            torch.cuda.current_stream().synchronize()  # in-flight chunks still read the old tables
            self._rot_tables(max_seqlen)

    def _build_desc(self):
        cfg = self.config
        L = cfg.n_layer
        self._layers_arr = (_C.CxLayerWeights * L)()
        P = lambda n: self.p(n).data_ptr() if n in self._layout else None  # noqa: E731
        G = lambda n: self.g(n).data_ptr() if n in self._layout else None  # noqa: E731
        for l in range(L):
            pre = self._LAYER_PREFIX.format(l=l)
            lw = self._layers_arr[l]
            for fld, nm in (("Wqkv", "attn.Wqkv.weight"), ("Wout", "attn.out_proj.weight"),
                            ("Wfc1", "mlp.fc1_fused.weight"), ("Wfc2", "mlp.fc2.weight")):
                setattr(lw, fld, self._w16(pre + nm).data_ptr())
                setattr(lw, fld + "T", self._w16(pre + nm, True).data_ptr())
                setattr(lw, "g" + fld, G(pre + nm))
            for fld, nm in (("bqkv", "attn.Wqkv.bias"), ("bout", "attn.out_proj.bias"),
                            ("bfc1", "mlp.fc1_fused.bias"), ("bfc2", "mlp.fc2.bias")):
                setattr(lw, fld, P(pre + nm))
                setattr(lw, "g" + fld, G(pre + nm))
            for fld, nm in (("ln1_g", "norm1.weight"), ("ln1_b", "norm1.bias"), ("ln2_g", "norm2.weight"),
                            ("ln2_b", "norm2.bias")):
                setattr(lw, fld, P(pre + nm))
                setattr(lw, "g" + fld, G(pre + nm))
        e = _C.CxEncoderDesc()
        e.n_layer, e.d, e.n_head, e.d_inner, e.gated = L, cfg.n_embd, cfg.n_head, cfg.n_inner, int(cfg.gated)
        e.vocab, e.padding_idx = getattr(cfg, "vocab_size", 0), getattr(cfg, "pad_token_id", 0)
        e.max_pos = (getattr(cfg, "max_position_embeddings", 0) if cfg.rotary_emb_fraction == 0
                     else getattr(cfg, "n_positions", 0))
        e.ln_eps = cfg.layer_norm_epsilon
        e.softmax_scale = 1.0 / math.sqrt(64.0)  # 1/norm_factor, sc/layers/attention.py:44-48,163
        e.word_emb = P("embeddings.word_embeddings.weight")
        e.type_emb = P("embeddings.token_type_embeddings.weight")
        e.pos_emb = P("embeddings.position_embeddings.weight")
        e.emb_ln_g, e.emb_ln_b = P("emb_ln.weight"), P("emb_ln.bias")
        e.gword_emb = G("embeddings.word_embeddings.weight")
        e.gtype_emb = G("embeddings.token_type_embeddings.weight")
        e.gpos_emb = G("embeddings.position_embeddings.weight")
        e.gemb_ln_g, e.gemb_ln_b = G("emb_ln.weight"), G("emb_ln.bias")
        e.rot_cos = None if self.rot_cos is None else self.rot_cos.data_ptr()
        e.rot_sin = None if self.rot_sin is None else self.rot_sin.data_ptr()
        e.layers = C.cast(self._layers_arr, C.POINTER(_C.CxLayerWeights))
        e.pool_mode, e.normalize = self.pool_mode, 1
        e.resid_pdrop, e.embd_pdrop = float(getattr(cfg, "resid_pdrop", 0.0)), float(getattr(cfg, "embd_pdrop", 0.0))
        e.attn_pdrop = float(getattr(cfg, "attn_pdrop", 0.0))
        self._desc = e

    # ------------------------------------------------------------------------------------------------ arenas
    ARENA_IDLE_USES = 16   # a saving arena nobody has taken for this many saving forwards of its engine is given back

    def _get_arena(self, T: int, B: int, save: bool) -> _ChunkArena:
        T_cap = _round_up(max(T, 1), 128)
        if save:
            # ragged batches: an arena serves every batch up to its capacity, so a loader's batches ratchet it up to the
            # largest one seen (~ln(steps) times per run); size classes (64 per octave, <= 1.6 % of slack) keep a record by
            # a handful of tokens from costing a rebuild, and the arenas left behind are given back below once idle
            T_cap = _round_up(T_cap, max(128, 1 << max(0, T_cap.bit_length() - 7)))
            self._arena_tick += 1
        if not save:
            a = self._arena_nograd
            if a is None or a.T_cap < T_cap or a.B_cap < B:
                a = _ChunkArena(self.config, T_cap, 1, False, max(B, 1), self.device_)
                self._arena_nograd = a
            return a
        ck = bool(self.gradient_checkpointing)
        # best fit: with two arena sizes in rotation (32 queries / 256 documents of cfg 3) a first-fit search hands the small
        # chunk the big arena and then has to build a second big one
        fits = [i for i, a in enumerate(self._arena_free) if a.T_cap >= T_cap and a.B_cap >= B and a.checkpoint == ck]
        if fits:
            a = self._arena_free.pop(min(fits, key=lambda i: self._arena_free[i].T_cap))
            a.last_tick = self._arena_tick
            return a
        # nothing fits: before building a bigger arena, give back the ones that have sat idle (a batch size that no longer
        # occurs; with selective checkpointing such an arena can hold most of the HBM).  Their plans are forgotten with them.
        stale = [a for a in self._arena_free if self._arena_tick - a.last_tick >= self.ARENA_IDLE_USES]
        if stale:
            self._arena_free = [a for a in self._arena_free if a not in stale]
            for a in stale:
                self._keep_plan.pop(a.T_cap, None)
            del stale, a
        L = self.config.n_layer
        if ck and self._keep_mode() == "auto" and not self._keep_suspended:
            # an arena that was measured and dropped for its rebuild is rebuilt at ITS size when a smaller batch comes first
            pending = [t for t in self._keep_granted if t >= T_cap]
            if pending:
                T_cap = min(pending)
                #  ... with the SEQUENCE capacity of the arena that was measured too: sized from the smaller batch that
                # happens to come first, the rebuilt arena would fail the `B_cap >= B` fit test when the batch it was planned for
                # arrives, and a second kept arena of the same T_cap would be built outside the ledger
                B = max(B, self._keep_granted_B.get(T_cap, 0))
        keep = self._checkpoint_keep_for(T_cap) if ck else 0
        if keep > 0:
            try:
                a = _ChunkArena(self.config, T_cap, L, True, max(B, 1), self.device_, checkpoint=True, keep_layers=keep)
                a.granted = self._keep_granted.pop(T_cap, 0)   # the ledger entry now belongs to the arena
                self._keep_granted_B.pop(T_cap, None)
                a.last_tick = self._arena_tick
                self._log_keep(T_cap, keep)
                return a
            except torch.OutOfMemoryError:
                # (fragmentation, another tenant of the device): nothing has been computed yet -> take the recipe literally
                torch.cuda.empty_cache()
                _hbm_grant(self.device_, -self._keep_granted.pop(T_cap, 0))
                self._keep_granted_B.pop(T_cap, None)
                if self._keep_mode() != "auto":
                    raise   # an explicit number is a demand, not a hint
                self._keep_plan[T_cap] = 0
                logging.getLogger("contrastors_amd").info(
                    "selective checkpointing: keeping %d of %d blocks ran out of memory at %d tokens; recomputing every block",
                    keep, L, T_cap)
        try:
            a = _ChunkArena(self.config, T_cap, L, True, max(B, 1), self.device_, checkpoint=ck)
        except torch.OutOfMemoryError:
            # a record batch while the arena it outgrew (with its kept blocks) still sits in the free list: every idle arena
            # goes back, whatever its age, and the allocation is tried once more
            if not self._arena_free:
                raise
            for old_arena in self._arena_free:
                self._keep_plan.pop(old_arena.T_cap, None)
            self._arena_free = []
            old_arena = None
            torch.cuda.empty_cache()
            a = _ChunkArena(self.config, T_cap, L, True, max(B, 1), self.device_, checkpoint=ck)
        a.probation = ck and self._keep_mode() == "auto" and not self._keep_suspended and T_cap not in self._keep_plan
        a.last_tick = self._arena_tick
        return a

    # ---- selective activation checkpointing (round 3): `gradient_checkpointing: true` is a memory knob sized for 80 GB ----
    # An integer `checkpoint_keep` is taken literally.  "auto" MEASURES instead of estimating: the first use of an arena takes
    # the recipe literally (every block recomputed); when its backward has run, the step's real peak is known -- other
    # towers' arenas, the loss, the optimizer included -- and the arena is rebuilt for its next use with as many kept
    # blocks as `CKPT_HBM_FRACTION` of the device minus that peak pays for.  A per-device ledger (`_hbm_grant`) keeps two
    # arenas (query / document side, two towers) from spending the same headroom twice.
    CKPT_HBM_FRACTION = 0.90

    def _keep_mode(self) -> Union[int, str]:
        return parse_checkpoint_keep(os.environ.get("CX_CHECKPOINT_KEEP"), "CX_CHECKPOINT_KEEP", default=self.checkpoint_keep)

    def _checkpoint_keep_for(self, T_cap: int) -> int:
        """Blocks of a NEW checkpointing arena that keep their intermediates (CxChunkBuffers.ckpt_keep)."""
        mode, L = self._keep_mode(), self.config.n_layer
        if mode != "auto":
            return max(0, min(int(mode), L))
        return 0 if self._keep_suspended else int(self._keep_plan.get(T_cap, 0))

    def _plan_keep(self, arena: _ChunkArena) -> bool:
        """Called when a probation arena's backward has been enqueued (release_arena): decide how many blocks its successor
        keeps.  True = the arena is to be dropped (its successor is built by the next saving forward)."""
        arena.probation = False
        if self._keep_mode() != "auto" or self._keep_suspended:
            return False
        L, T_cap = self.config.n_layer, arena.T_cap
        per_keep = T_cap * _ChunkArena.slot_bytes_per_token(self.config)
        total = torch.cuda.get_device_properties(self.device_).total_memory
        peak = torch.cuda.max_memory_allocated(self.device_)
        budget = self.CKPT_HBM_FRACTION * total - peak - _hbm_grant(self.device_, 0)
        # ... and never more than the device has free right now (another process on the same GPU does not show up in this
        # process's peak): driver-level free memory + the allocator's cache + the arena that is about to be dropped
        free_now, _ = torch.cuda.mem_get_info(self.device_)
        free_now += torch.cuda.memory_reserved(self.device_) - torch.cuda.memory_allocated(self.device_) + arena.nbytes()
        budget = min(budget, self.CKPT_HBM_FRACTION * free_now - arena.nbytes() - _hbm_grant(self.device_, 0))
        keep = int(max(0, min(L, budget // max(per_keep, 1))))
        self._keep_plan[T_cap] = keep
        if keep == 0:
            return False
        _hbm_grant(self.device_, keep * per_keep)
        self._keep_granted[T_cap] = keep * per_keep   # handed to the successor arena when it is built
        self._keep_granted_B[T_cap] = int(getattr(arena, "B_cap", 0))
        return True

    @contextlib.contextmanager
    def selective_checkpointing_suspended(self):
        """A schedule that keeps many arenas alive at once (GradCache with resident activations: one per chunk) has already
        budgeted the HBM for the recipe's literal checkpointing; `auto` stays out of its way."""
        self._keep_suspended += 1
        try:
            yield
        finally:
            self._keep_suspended -= 1

    def _log_keep(self, T_cap: int, keep: int):
        key = (T_cap, keep)
        if key in self._keep_logged:
            return
        self._keep_logged.add(key)
        L = self.config.n_layer
        logging.getLogger("contrastors_amd").info(
            "selective checkpointing: %d-token arena keeps the activations of %d of %d blocks (%.1f GB), %d are recomputed "
            "in backward (checkpoint_keep_layers = %r)", T_cap, keep, L,
            keep * T_cap * _ChunkArena.slot_bytes_per_token(self.config) / 1e9, L - keep, self.checkpoint_keep)

    def _arm_dropout(self, arena: _ChunkArena):
        """Dropout is active in training mode when the config asks for it.  The Philox (seed, offset) of the chunk comes
        from torch's generator of this device and advances it, exactly like a torch dropout op would: RandContext
        (rand_state.py; sc/rand_state.py:6-22) snapshots and restores that state around the GradCache re-forward, so the
        second forward of a chunk regenerates the first one's masks."""
        cfg = self.config
        active = self.training and any(getattr(cfg, k, 0.0) > 0 for k in ("resid_pdrop", "embd_pdrop", "attn_pdrop"))
        arena.desc.drop_active = int(active)
        if active:
            gen = torch.cuda.default_generators[self.device_.index if self.device_.index is not None else torch.cuda.current_device()]
            off = gen.get_offset()
            # the kernels key dropout site s of this chunk as Philox counter word (offset + s) (cx_common.h dropout_keep4);
            # a chunk has 3 L + 1 sites (2l, 2l + 1 residual, 2L embeddings, 2L + 1 + l attention), so the generator moves
            # past ALL of them -- advancing by less would hand the next chunk this chunk's streams shifted by a few sites
            # (torch keeps Philox offsets in multiples of 4)
            gen.set_offset(off + self.dropout_offset_stride(cfg.n_layer))
            arena.desc.drop_seed, arena.desc.drop_offset = gen.initial_seed() & (2**64 - 1), off

    @staticmethod
    def dropout_offset_stride(n_layer: int) -> int:
        """Philox offsets one chunk consumes: its 3 L + 1 dropout sites rounded up to torch's granularity of 4."""
        return 4 * ((3 * n_layer + 1 + 3) // 4)

    @property
    def uses_rng(self) -> bool:
        cfg = self.config
        return self.training and any(getattr(cfg, k, 0.0) > 0 for k in ("resid_pdrop", "embd_pdrop", "attn_pdrop"))

    def release_arena(self, arena: _ChunkArena, used: bool = True):
        arena.emb_out = None
        arena.desc.layer_events = None
        if used and getattr(arena, "probation", False) and self._plan_keep(arena):
            return   # dropped: the next saving forward builds its successor with the planned number of kept blocks
        self._arena_free.append(arena)

    def abandon_arena(self, arena: _ChunkArena):
        """A saved forward whose backward will never run (grad_cache_loss falling back after an out-of-memory error)."""
        self._outstanding = max(0, self._outstanding - 1)
        self.release_arena(arena, used=False)

    # ---- data-parallel gradient reduction overlapped with the step's last backward (what DDP's bucket hooks do for the
    #      reference, sc/trainers/text_text.py:163-170) ------------------------------------------------
    def grad_ranges(self):
        """(per-block ranges, tail ranges) of the flat gradient buffer: block l owns one contiguous run of the decay group
        (its four Linear weights) and one of the no-decay group (biases, LayerNorms); the tail is everything else
        (embeddings, emb_ln / cls, pos, patch projection, ln_f)."""
        L = self.config.n_layer
        per_layer, covered = [], []
        for l in range(L):
            pre = self._LAYER_PREFIX.format(l=l)
            runs = []
            for lo, hi in ((0, self.n_decay), (self.n_decay, self.n_total)):
                offs = [(off, off + _round_up(int(np.prod(shape)), 64)) for n, (off, shape) in self._layout.items()
                        if n.startswith(pre) and lo <= off < hi]
                if offs:
                    a, b = min(o[0] for o in offs), max(o[1] for o in offs)
                    assert sum(o[1] - o[0] for o in offs) == b - a, "a block's parameters are contiguous per group"
                    runs.append((a, b))
            per_layer.append(runs)
            covered += runs
        covered.sort()
        tail, pos = [], 0
        for a, b in covered + [(self.n_total, self.n_total)]:
            if a > pos:
                tail.append((pos, a))
            pos = max(pos, b)
        return per_layer, tail

    def arm_overlapped_reduce(self, when_last_outstanding: bool = False):
        """The next backward_chunk (or, with `when_last_outstanding`, the one that consumes the last saved forward) is the
        final contribution to this step's gradients: have it record one event per block and start the all-reduce of each
        block's slice on a side stream as soon as the block's gradients exist.  No-op for a single process.
        `finish_overlapped_reduce()` (BiEncoder.sync_gradients) waits for the collectives and applies the 1 / W."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        self._overlap_armed = "last" if when_last_outstanding else "next"

    def _overlap_fires(self) -> bool:
        armed = getattr(self, "_overlap_armed", None)
        if armed == "next" or (armed == "last" and self._outstanding == 1):
            self._overlap_armed = None
            return True
        return False

    def _overlap_events(self):
        L = self.config.n_layer
        if getattr(self, "_ov_events", None) is None:
            self._ov_events = [torch.cuda.Event() for _ in range(L + 1)]
            for e in self._ov_events:
                e.record(torch.cuda.current_stream(self.device_))   # materialises the hipEvent_t behind the object
            self._ov_handles = (C.c_void_p * (L + 1))(*[e.cuda_event for e in self._ov_events])
            self._ov_stream = torch.cuda.Stream(device=self.device_)
            self._ov_ranges = self.grad_ranges()
        return self._ov_events

    def _launch_overlapped_reduce(self):
        """Called right after the final backward has been ENQUEUED: the side stream waits for each block's event and issues
        that block's all-reduces; the compute stream is not blocked."""
        ev = self._ov_events
        per_layer, tail = self._ov_ranges
        works = []
        with torch.cuda.stream(self._ov_stream):
            for l in range(self.config.n_layer - 1, -1, -1):
                self._ov_stream.wait_event(ev[l])
                for a, b in per_layer[l]:
                    works.append(dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True))
            self._ov_stream.wait_event(ev[self.config.n_layer])
            for a, b in tail:
                works.append(dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True))
        self._ov_works = works

    def finish_overlapped_reduce(self) -> bool:
        """True when an overlapped reduction was in flight: the current stream now waits for it (SUM over ranks is in
        flat_grad; the caller divides by W).  False: nothing was launched, the caller reduces the flat buffer itself."""
        works = getattr(self, "_ov_works", None)
        if not works:
            return False
        for w in works:
            w.wait()
        torch.cuda.current_stream(self.device_).wait_stream(self._ov_stream)
        self._ov_works = None
        return True

    def gradient_checkpointing_enable(self, enabled: bool = True, keep_layers: Union[int, str, None] = None):
        """`keep_layers` (beyond the reference's signature): how many blocks keep their activations anyway -- an integer,
        or "auto" = as many as the free HBM takes (selective checkpointing; results are bit-identical for every value)."""
        self.gradient_checkpointing = bool(enabled)
        if keep_layers is not None:
            self.checkpoint_keep = parse_checkpoint_keep(keep_layers, "checkpoint_keep_layers")
        self._arena_free = []   # (arenas of the other mode, or built for another keep count, are not reused)

    # ------------------------------------------------------------------------------------------------ compute
    def forward_chunk(self, vb: VarlenBatch, save_for_backward: bool, normalize: Optional[bool] = None,
                      out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[_ChunkArena]]:
        """Enqueue one chunk forward.  Returns (embeddings (B,d) fp32, arena holding the saved activations)."""
        d = self.config.n_embd
        if self.config.rotary_emb_fraction > 0 and self.config.rotary_scaling_factor:
            self._update_rotary(vb.max_seqlen)
        elif vb.S > (self.config.n_positions if self.rot_cos is not None else self.config.max_position_embeddings):
            raise ValueError("sequence longer than the position table")
        if out is None:
            out = torch.empty(vb.B, d, dtype=torch.float32, device=self.device_)
        arena = self._get_arena(vb.T, vb.B, save_for_backward)
        self._arm_dropout(arena)
        self._desc.normalize = int(self.normalize_default if normalize is None else normalize)
        rc = self.lib.cx_encoder_forward(C.byref(self._desc), C.byref(arena.desc), vb.input_ids.data_ptr(),
                                         vb.indices.data_ptr(), vb.cu_seqlens.data_ptr(), vb.B, vb.S, vb.T,
                                         vb.max_seqlen, int(save_for_backward), out.data_ptr(), _C.cur_stream())
        _C.check(rc, "cx_encoder_forward")
        if save_for_backward:
            arena.emb_out = out
            arena.normalize = self._desc.normalize
            self._outstanding += 1
            return out, arena
        return out, None

    def drop_idle_arenas(self):
        """Give every pooled (idle) activation arena back to the allocator: a schedule change (another batch shape, another
        GradCache policy) then starts from exactly-sized arenas instead of best-fitting into the previous schedule's."""
        for a in self._arena_free:
            self._keep_plan.pop(a.T_cap, None)
        self._arena_free = []
        self._arena_nograd = None
        torch.cuda.empty_cache()

    def backward_chunk(self, vb: VarlenBatch, arena: _ChunkArena, demb: torch.Tensor):
        """Accumulate every parameter gradient for the chunk whose activations are in `arena`."""
        assert arena.emb_out is not None, "backward_chunk needs a forward with save_for_backward=True"
        demb = demb.to(torch.float32).contiguous()
        self._desc.normalize = arena.normalize
        sids, perm = vb.embedding_sort()
        fires = self._begin_backward(arena)
        rc = self.lib.cx_encoder_backward(C.byref(self._desc), C.byref(arena.desc), vb.input_ids.data_ptr(),
                                          vb.indices.data_ptr(), vb.cu_seqlens.data_ptr(), vb.B, vb.S, vb.T,
                                          vb.max_seqlen, demb.data_ptr(), arena.emb_out.data_ptr(), _C.ptr(sids),
                                          _C.ptr(perm), _C.cur_stream())
        _C.check(rc, "cx_encoder_backward")
        self._end_backward(arena, fires)

    def _begin_backward(self, arena: _ChunkArena) -> bool:
        fires = self._overlap_fires()
        if fires:
            self._overlap_events()
            arena.desc.layer_events = C.cast(self._ov_handles, C.POINTER(C.c_void_p))
        return fires

    def _end_backward(self, arena: _ChunkArena, fires: bool):
        self._outstanding = max(0, self._outstanding - 1)
        if fires:
            self._launch_overlapped_reduce()
        self.release_arena(arena)

    # ---- token-level outputs (MLM head): (T, d) bf16 hidden states in unpadded order -----------------------------
    def _rotary_or_bounds(self, vb: VarlenBatch):
        if self.config.rotary_emb_fraction > 0 and self.config.rotary_scaling_factor:
            self._update_rotary(vb.max_seqlen)
        elif vb.S > (self.config.n_positions if self.rot_cos is not None else self.config.max_position_embeddings):
            raise ValueError("sequence longer than the position table")

    def forward_hidden_chunk(self, vb: VarlenBatch, save_for_backward: bool):
        self._rotary_or_bounds(vb)
        hidden = torch.empty(vb.T, self.config.n_embd, dtype=torch.bfloat16, device=self.device_)
        arena = self._get_arena(vb.T, vb.B, save_for_backward)
        self._arm_dropout(arena)
        rc = self.lib.cx_encoder_forward_hidden(C.byref(self._desc), C.byref(arena.desc), vb.input_ids.data_ptr(),
                                                vb.indices.data_ptr(), vb.cu_seqlens.data_ptr(), vb.B, vb.S, vb.T,
                                                vb.max_seqlen, int(save_for_backward), hidden.data_ptr(),
                                                _C.cur_stream())
        _C.check(rc, "cx_encoder_forward_hidden")
        return hidden, (arena if save_for_backward else None)

    def backward_hidden_chunk(self, vb: VarlenBatch, arena: _ChunkArena, dhidden: torch.Tensor):
        dh = dhidden.to(torch.bfloat16).contiguous()
        assert dh.shape == (vb.T, self.config.n_embd)
        sids, perm = vb.embedding_sort()
        rc = self.lib.cx_encoder_backward_hidden(C.byref(self._desc), C.byref(arena.desc), vb.input_ids.data_ptr(),
                                                 vb.indices.data_ptr(), vb.cu_seqlens.data_ptr(), vb.B, vb.S, vb.T,
                                                 vb.max_seqlen, dh.data_ptr(), _C.ptr(sids), _C.ptr(perm),
                                                 _C.cur_stream())
        _C.check(rc, "cx_encoder_backward_hidden")
        self.release_arena(arena)

    def hidden_states(self, vb: VarlenBatch) -> torch.Tensor:
        """(T, d) bf16 last hidden state of the unpadded tokens; differentiable w.r.t. the engine's parameters."""
        if torch.is_grad_enabled() and self.training:
            return _HiddenFn.apply(self.flat_decay, self, vb)
        return self.forward_hidden_chunk(vb, False)[0]

    # nn.Module-style call used by BiEncoder: differentiable w.r.t. the engine's own parameters
    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                normalize: Optional[bool] = None) -> torch.Tensor:
        vb = VarlenBatch.from_mask(input_ids, attention_mask)
        if torch.is_grad_enabled() and self.training:
            return _EncodeFn.apply(self.flat_decay, self, vb, normalize)
        emb, _ = self.forward_chunk(vb, False, normalize)
        return emb


class _EncodeFn(torch.autograd.Function):
    """autograd bridge: backward accumulates straight into the engine's flat gradient buffer (as the reference's
    `surrogate.backward()` accumulates into .grad, sc/loss.py:158-161) and reports no gradient for its inputs."""

    @staticmethod
    def forward(ctx, _anchor, engine: NomicBertEngine, vb: VarlenBatch, normalize):
        emb, arena = engine.forward_chunk(vb, True, normalize)
        ctx.engine, ctx.vb, ctx.arena = engine, vb, arena
        return emb

    @staticmethod
    def backward(ctx, demb):
        arena, ctx.arena = ctx.arena, None   # (a graph that outlives its backward must not keep the arena alive)
        ctx.engine.backward_chunk(ctx.vb, arena, demb)
        return None, None, None, None


class _HiddenFn(torch.autograd.Function):
    """Token-level twin of _EncodeFn: hidden states out, d(hidden) in, parameter gradients into the flat buffer."""

    @staticmethod
    def forward(ctx, _anchor, engine: NomicBertEngine, vb: VarlenBatch):
        hidden, arena = engine.forward_hidden_chunk(vb, True)
        ctx.engine, ctx.vb, ctx.arena = engine, vb, arena
        return hidden

    @staticmethod
    def backward(ctx, dhidden):
        arena, ctx.arena = ctx.arena, None
        ctx.engine.backward_hidden_chunk(ctx.vb, arena, dhidden)
        return None, None, None
