"""Masked-language-model pretraining on the native trunk: host-side mirror of
`NomicBertForPreTraining` (sc/models/encoder/modeling_nomic_bert.py:590-669, heads :416-483) and `MLMTrainer`
(sc/trainers/mlm.py:16-153).

One encoder call (`cx_encoder_forward_hidden`) yields the (T, d) hidden states of the unpadded tokens; with
`dense_seq_output` only the rows that carry a label (~15-30 %) go through the head: transform (Linear -> SiLU / GELU ->
LayerNorm), the vocabulary projection tied to the word-embedding matrix (bf16 MFMA GEMM, 30528-way), and the fused
cross-entropy kernel with in-place backward (K12) -- the (M, V) logits are written once and overwritten by their own
gradient.  State-dict keys are the reference's (`bert.*`, `cls.predictions.transform.*`, `cls.predictions.decoder.bias`;
the decoder weight is the embedding matrix and is not stored twice).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist

from .config import Config
from .flash_attn_api.losses.cross_entropy import CrossEntropyLoss
from .flash_attn_api.ops.fused_dense import fused_dense_func
from .flash_attn_api.ops.layer_norm import layer_norm
from .nomic_bert import NomicBertConfig, NomicBertEngine, VarlenBatch
from .optimizer import FusedAdamW

_WORD = "embeddings.word_embeddings.weight"


class NomicBertForPreTraining(torch.nn.Module):
    def __init__(self, config: NomicBertConfig, device="cuda", seed: Optional[int] = None, dense_seq_output: bool = True):
        super().__init__()
        self.config = config
        self.dense_seq_output = dense_seq_output
        self.bert = NomicBertEngine(config, device=device, pooling="mean", normalize=False, seed=seed)
        d, V = config.n_embd, config.vocab_size
        dev = self.bert.device_
        gen = torch.Generator(device="cpu").manual_seed(1 if seed is None else seed + 1)
        # sc/models/encoder/modeling_nomic_bert.py:416-470: dense(d, d, bias=mlp_fc1_bias), act, LayerNorm, decoder bias
        self.dense_weight = torch.nn.Parameter(
            torch.empty(d, d).normal_(0.0, config.initializer_range, generator=gen).to(dev))
        self.dense_bias = torch.nn.Parameter(torch.zeros(d, device=dev)) if config.mlp_fc1_bias else None
        self.ln_weight = torch.nn.Parameter(torch.ones(d, device=dev))
        self.ln_bias = torch.nn.Parameter(torch.zeros(d, device=dev))
        self.decoder_bias = torch.nn.Parameter(torch.zeros(V, device=dev)) if config.mlp_fc1_bias else None
        self.mlm_loss = CrossEntropyLoss(inplace_backward=True)  # :603-605 (use_xentropy)
        self._tied = None

    # ---- the tied decoder weight: an autograd leaf over the engine's embedding matrix; its gradient is folded into the
    #      engine's flat gradient buffer by `fold_tied_grad` (what weight tying does for the reference, :613-615)
    def _tied_weight(self) -> torch.Tensor:
        w = self.bert.p(_WORD).detach()
        w.requires_grad_(self.training and torch.is_grad_enabled())
        self._tied = w
        return w

    def fold_tied_grad(self):
        if self._tied is not None and self._tied.grad is not None:
            self.bert.g(_WORD).add_(self._tied.grad)
            self._tied.grad = None
        self._tied = None

    def _act(self, x):
        if self.config.activation_function == "swiglu":
            return torch.nn.functional.silu(x)
        approx = "tanh" if self.config.activation_function in ("gelu_new", "gelu_fast", "gelu_pytorch_tanh") else "none"
        return torch.nn.functional.gelu(x, approximate=approx)

    def head(self, hidden: torch.Tensor) -> torch.Tensor:
        h = fused_dense_func(hidden, self.dense_weight, self.dense_bias)
        h = self._act(h)
        h = layer_norm(h, self.ln_weight, self.ln_bias, self.config.layer_norm_epsilon)
        return fused_dense_func(h, self._tied_weight(), self.decoder_bias)

    def forward(self, input_ids, position_ids=None, token_type_ids=None, attention_mask=None, labels=None):
        """labels: (B, S) int64, -100 on every position that is not a prediction target (padding included)."""
        if position_ids is not None or token_type_ids is not None:
            raise NotImplementedError("explicit position / token-type ids (the MLM recipe passes neither)")
        dev = self.bert.device_
        input_ids = input_ids.to(dev)
        vb = VarlenBatch.from_mask(input_ids, None if attention_mask is None else attention_mask.to(dev))
        hidden = self.bert.hidden_states(vb)  # (T, d) bf16, unpadded order
        if labels is None:
            logits = self.head(hidden)
            full = torch.zeros(vb.B * vb.S, logits.shape[-1], dtype=logits.dtype, device=dev)
            full[vb.indices.long()] = logits
            return SimpleNamespace(loss=None, prediction_logits=full.view(vb.B, vb.S, -1))
        lab = labels.to(dev).flatten()[vb.indices.long()]
        if self.dense_seq_output:
            idx = torch.nonzero(lab >= 0, as_tuple=False).flatten()  # :650-653 masked_token_idx
            hidden, lab = hidden.index_select(0, idx), lab[idx]
        logits = self.head(hidden)
        loss = self.mlm_loss(logits, lab).float()
        return SimpleNamespace(loss=loss, prediction_logits=logits)

    # ---- parameters ----------------------------------------------------------------------------------------------------
    def head_parameters(self):
        return [p for p in (self.dense_weight, self.dense_bias, self.ln_weight, self.ln_bias, self.decoder_bias)
                if p is not None]

    def param_groups(self, weight_decay: float):
        """sc/optimizer.py:16-25: matrices decay, biases / LayerNorm do not."""
        return [{"params": [self.bert.flat_decay, self.dense_weight], "weight_decay": weight_decay},
                {"params": [self.bert.flat_nodecay] + [p for p in self.head_parameters() if p is not self.dense_weight],
                 "weight_decay": 0.0}]

    def zero_grad(self, set_to_none: bool = False):
        self.bert.zero_grad()
        for p in self.head_parameters():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def sync_gradients(self):
        """DDP's job in the reference (sc/trainers/mlm.py:47-51): average every gradient over ranks, once per step."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            W = dist.get_world_size()
            dist.all_reduce(self.bert.flat_grad)
            self.bert.flat_grad.div_(W)
            for p in self.head_parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad.div_(W)

    def reference_state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {f"bert.{k}": v for k, v in self.bert.reference_state_dict().items()}
        sd["cls.predictions.transform.dense.weight"] = self.dense_weight.detach()
        sd["cls.predictions.transform.layer_norm.weight"] = self.ln_weight.detach()
        sd["cls.predictions.transform.layer_norm.bias"] = self.ln_bias.detach()
        if self.dense_bias is not None:
            sd["cls.predictions.transform.dense.bias"] = self.dense_bias.detach()
            sd["cls.predictions.decoder.bias"] = self.decoder_bias.detach()
        sd["cls.predictions.decoder.weight"] = sd["bert." + _WORD]  # tied (:613-615)
        return sd

    @torch.no_grad()
    def load_reference_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        self.bert.load_reference_state_dict({k[5:]: v for k, v in sd.items() if k.startswith("bert.")}, strict=strict)
        pairs = [("cls.predictions.transform.dense.weight", self.dense_weight),
                 ("cls.predictions.transform.dense.bias", self.dense_bias),
                 ("cls.predictions.transform.layer_norm.weight", self.ln_weight),
                 ("cls.predictions.transform.layer_norm.bias", self.ln_bias),
                 ("cls.predictions.decoder.bias", self.decoder_bias)]
        for k, p in pairs:
            if p is None:
                continue
            if k in sd:
                p.copy_(sd[k].to(p.device, torch.float32))
            elif strict:
                raise KeyError(k)


def mask_tokens(input_ids: torch.Tensor, special_tokens_mask: torch.Tensor, mlm_probability: float, mask_token_id: int,
                vocab_size: int, generator: Optional[torch.Generator] = None):
    """transformers.DataCollatorForLanguageModeling.torch_mask_tokens, the collator sc/trainers/mlm.py:70,84 uses:
    each non-special token is a target with probability p; of the targets 80 % become [MASK], 10 % a random token,
    10 % stay.  Returns (masked input_ids, labels with -100 off-target)."""
    labels = input_ids.clone()
    prob = torch.full(labels.shape, mlm_probability)
    prob.masked_fill_(special_tokens_mask.bool(), 0.0)
    target = torch.bernoulli(prob, generator=generator).bool()
    labels[~target] = -100
    ids = input_ids.clone()
    replaced = torch.bernoulli(torch.full(labels.shape, 0.8), generator=generator).bool() & target
    ids[replaced] = mask_token_id
    rand = torch.bernoulli(torch.full(labels.shape, 0.5), generator=generator).bool() & target & ~replaced
    words = torch.randint(vocab_size, labels.shape, dtype=torch.long, generator=generator)
    ids[rand] = words[rand]
    return ids, labels


class MLMTrainer:
    """sc/trainers/mlm.py on the native path: forward_step = model(**batch).loss; training_step =
    base.py:366-393 with gradient accumulation (mlm.yaml uses 4), fused clip + AdamW, linear / cosine schedule."""

    def __init__(self, config: Config, dtype=torch.bfloat16, device=None, trunk_config: Optional[NomicBertConfig] = None,
                 total_steps: Optional[int] = None):
        if dtype != torch.bfloat16:
            raise NotImplementedError("the native path computes in bf16 with fp32 master weights (--dtype=bf16)")
        from .trainers import _lr_lambda

        self.config = config
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        ta, ma = config.train_args, config.model_args
        torch.manual_seed(config.data_args.seed)
        if trunk_config is None:
            # sc/trainers/mlm.py:20-40: bert-base-uncased geometry (no hub access: its config constants are spelled out)
            # with the recipe's overrides, then bert_config_to_nomic_config (sc/models/encoder/bert.py:11-50)
            mult = ma.pad_vocab_to_multiple_of or 1
            vocab = (30522 + mult - 1) // mult * mult  # modeling_nomic_bert.py:488-490 pads the vocabulary
            if ma.use_rms_norm:
                raise NotImplementedError("RMSNorm trunks")
            trunk_config = NomicBertConfig(
                vocab_size=vocab, n_positions=ma.seq_len, max_position_embeddings=ma.seq_len,
                activation_function=ma.activation_function or "gelu", rotary_emb_fraction=ma.rotary_emb_fraction or 0.0,
                rotary_emb_base=ma.rotary_emb_base or 10_000, qkv_proj_bias=bool(ma.qkv_proj_bias),
                mlp_fc1_bias=bool(ma.mlp_fc1_bias), mlp_fc2_bias=bool(ma.mlp_fc2_bias), attn_pdrop=ma.attn_pdrop or 0.0,
                # bert-base-uncased's hidden_dropout_prob = 0.1 becomes resid_pdrop and embd_pdrop (bert.py:20-21); the
                # recipe only overrides the attention dropout (trainers/mlm.py:37)
                resid_pdrop=0.1, embd_pdrop=0.1,
                layer_norm_epsilon=1e-12, type_vocab_size=2, pad_token_id=0)
        model = NomicBertForPreTraining(trunk_config, device=self.device, seed=config.data_args.seed).train()
        if self.world > 1:
            dist.broadcast(model.bert.flat_param, 0)
            for p in model.head_parameters():
                dist.broadcast(p.data, 0)
            model.bert.sync_shadows()
        self.model = {"model": model}
        self.total_steps = total_steps or ta.num_train_steps or 10_000
        self.optimizer = FusedAdamW(model.param_groups(ta.weight_decay), lr=ta.learning_rate,
                                    betas=(ta.adam_beta1, ta.adam_beta2), eps=ta.eps)
        if ta.warmup_steps is None and ta.warmup_pct is None:
            raise ValueError("warmup_steps or warmup_pct must be set")
        warm = ta.warmup_steps if ta.warmup_steps is not None else int(ta.warmup_pct * self.total_steps)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, _lr_lambda(ta.schedule_type, warm, self.total_steps))
        self.accum = max(1, int(getattr(ta, "gradient_accumulation_steps", 1) or 1))
        self.step = 0

    def forward_step(self, batch) -> torch.Tensor:
        model = self.model["model"]
        return model(**{k: v for k, v in batch.items() if k in ("input_ids", "attention_mask", "labels")}).loss

    def training_step(self, batch) -> torch.Tensor:
        """One micro-batch (base.py:366-393).  Reference quirks kept: `backward` is a plain `loss.backward()` (:346-351,
        accumulated gradients are summed, not averaged) and with accumulation the clip fires on micro-step
        `step % accum == 0` while the optimizer fires on `(step + 1) % accum == 0` (:375-385); with
        accum == 1 (every shipped contrastive recipe) clip + AdamW are the fused kernels."""
        ta, model = self.config.train_args, self.model["model"]
        loss = self.forward_step(batch)
        loss.backward()
        model.fold_tied_grad()
        clip = ta.max_grad_norm is not None and ta.max_grad_norm > 0
        fire = (self.step + 1) % self.accum == 0 or self.step == self.total_steps - 1   # (base.py:381: the run's last micro-step steps too)
        if fire:
            model.sync_gradients()  # DDP reduces on the micro-step that ends the window
        if clip and self.accum > 1 and self.step % self.accum == 0:
            params = [p for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
            torch.nn.utils.clip_grad_norm_(params, ta.max_grad_norm)
        if fire:
            self.optimizer.step(max_grad_norm=ta.max_grad_norm if (clip and self.accum == 1) else None)
            self.scheduler.step()
            model.bert.sync_shadows()
            model.zero_grad(set_to_none=False)
        self.step += 1
        return loss.detach()

    def train(self, batches: Iterable[dict], max_steps: Optional[int] = None, log_every: int = 0):
        """Same driver loop as TextTextTrainer.train (one call = one micro-batch)."""
        losses = []
        rank = dist.get_rank() if self.distributed else 0
        for i, batch in enumerate(batches):
            if max_steps is not None and i >= max_steps:
                break
            loss = self.training_step(batch)
            losses.append(loss)
            if log_every and (i + 1) % log_every == 0 and rank == 0:
                print(f"step {self.step} loss {float(loss):.4f} lr {self.scheduler.get_last_lr()[0]:.3e}", flush=True)
        return losses

    @torch.no_grad()
    def eval_step(self, batch) -> torch.Tensor:
        model = self.model["model"]
        was = model.training
        model.eval()
        try:
            return self.forward_step(batch).detach()
        finally:
            model.train(was)


def synthetic_mlm_batches(n_steps: int, batch: int, seq_len: int, vocab: int = 30522, mlm_probability: float = 0.3,
                          mask_token_id: int = 103, seed: int = 1234, ragged: bool = True) -> Iterable[dict]:
    g = torch.Generator().manual_seed(seed)
    for _ in range(n_steps):
        ids = torch.randint(min(1000, vocab // 2), vocab, (batch, seq_len), generator=g)
        lens = torch.randint(seq_len // 2, seq_len + 1, (batch,), generator=g) if ragged else torch.full((batch,), seq_len)
        mask = (torch.arange(seq_len)[None, :] < lens[:, None]).long()
        ids = ids * mask
        ids[:, 0] = min(101, vocab - 1)
        special = (mask == 0) | (torch.arange(seq_len)[None, :] == 0)
        mids, labels = mask_tokens(ids, special, mlm_probability, mask_token_id, vocab, g)
        yield {"input_ids": mids, "attention_mask": mask, "labels": labels}
