"""TextTextTrainer / ImageTextTrainer on the native path (host-side mirror of sc/trainers/base.py:203-208,354-533 and
sc/trainers/text_text.py:139-182,276-322,429-451 for the GradCache contrastive recipe).

Kept: the method set (`get_model`, `get_optimizer`, `get_scheduler`, `forward_step`, `backward`, `training_step`,
`train`), the batch contract of the streaming loader (sc/dataset/text_text_loader.py:601-660: `query_input_ids`,
`query_attention_mask`, `document_input_ids`, `document_attention_mask`, optional `dataset_name`) and the step order
(clip -> optimizer -> scheduler -> zero_grad).  Not rebuilt: S3 shard streaming, wandb, NanoBEIR eval, checkpoints
(out of the hot path): `train()` consumes any iterable of batches.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Iterable, Optional

import numpy as np
import torch
import torch.distributed as dist

from .biencoder import BiEncoder, BiEncoderConfig, DualEncoder, LogitScale
from .config import Config
from .distributed import check_exchange, gather_with_grad, set_exchange_mode, set_exchange_timeout
from .loss import clip_loss, grad_cache_loss
from .policy import GradCachePolicy, _parse_keep
from .nomic_bert import NomicBertConfig
from .optimizer import FusedAdamW


def _lr_lambda(schedule: str, warmup: int, total: int):
    """The multiplier `transformers.get_scheduler(name=schedule_type, optimizer, num_warmup_steps, num_training_steps)` puts on
    the base learning rate at scheduler step `step` -- the scheduler the reference builds (sc/trainers/base.py:258-263) --
    restated so that the trainers do not depend on the library; pinned to it step by step in tests/test_host_cpu.py.
    Note what that means: the warm-up starts AT ZERO (the first optimizer step of a run with warmup_steps > 0 moves nothing)
    and reaches the base rate at step == warmup, and the cosine is not clamped past `total`.  (Rounds 1-3 had the warm-up one
    step early: (step + 1) / warmup.)"""
    if schedule not in ("cosine", "linear", "constant", "constant_with_warmup", "inverse_sqrt"):
        raise ValueError(f"schedule_type {schedule!r} is not served (cosine, linear, constant, constant_with_warmup, inverse_sqrt)")

    def f(step: int) -> float:
        if schedule == "constant":
            return 1.0
        if step < warmup:
            return float(step) / float(max(1, warmup))
        if schedule == "constant_with_warmup":
            return 1.0
        if schedule == "inverse_sqrt":   # (sc/trainers/base.py:262 passes no horizon; timescale = warmup or 10000)
            timescale = warmup or 10_000
            return 1.0 / math.sqrt((step + timescale - warmup) / timescale)
        if schedule == "linear":
            return max(0.0, float(total - step) / float(max(1, total - warmup)))
        progress = float(step - warmup) / float(max(1, total - warmup))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))

    return f


def _load_initial_weights(model: BiEncoder, ma, explicit_arch: bool):
    """sc/trainers/text_text.py:139-160 + sc/models/biencoder/modeling_biencoder.py:160-250: `checkpoint` = a
    BiEncoder.save_pretrained directory, `pretrained` = hub weights of `model_name`.  There is no hub here: a local
    directory in `model_name` is loaded (native safetensors layout, or a HuggingFace BERT checkpoint through the
    reference's remap, hf_bert.py); otherwise `pretrained: true` RAISES instead of silently training from a random init.
    An explicitly passed trunk architecture (tests, benchmarks) is a declared random init."""
    import os

    ckpt = getattr(ma, "checkpoint", None)
    if ckpt:
        model.load_pretrained(ckpt)
        return
    if not getattr(ma, "pretrained", False) or explicit_arch:
        return
    name = ma.model_name or ""
    if os.path.isdir(name):
        if os.path.exists(os.path.join(name, "model.safetensors")) and os.path.exists(os.path.join(name, "config.json")):
            import json

            cfg = json.load(open(os.path.join(name, "config.json")))
            if "trunk_config" in cfg:  # written by BiEncoder.save_pretrained
                model.load_pretrained(name)
                return
        from transformers import BertConfig

        from .hf_bert import load_hf_bert

        hf_cfg = BertConfig.from_pretrained(name, local_files_only=True)
        st = os.path.join(name, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file

            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(name, "pytorch_model.bin"), map_location="cpu")
        load_hf_bert(model.trunk, sd, hf_cfg)
        return
    raise FileNotFoundError(
        f"model_args.pretrained is true but {name!r} is not a local directory (no hub access): point model_name / "
        "checkpoint at local weights, or set pretrained: false to train from a random init")


def _accumulation_steps(ta) -> int:
    """gradient_accumulation_steps of the recipe (sc/trainers/base.py:366-393).  Every contrastive recipe of the reference
    leaves it at 1 (the global batch is what GradCache is for); > 1 runs the reference's micro-step schedule below."""
    n = getattr(ta, "gradient_accumulation_steps", 1)
    n = 1 if n is None else int(n)
    if n < 1:
        raise ValueError(f"gradient_accumulation_steps must be >= 1, got {n}")
    return n


# Below this many token rows per side a direct step's two tower calls (sc/trainers/text_text.py:330-345: model(query),
# model(document)) are ONE call on the concatenated batch: every launch of BASELINE configs[0] (B = 32, S = 64) is 2048
# token rows = 8 row panels of 256 on 256 CUs, its duration is latency, and two such launches back to back take twice as
# long as one of 4096 rows.  Every op of the trunk is per token or per sequence, so embeddings are the same numbers; the
# weight gradients sum the two sides' tokens in one reduction instead of two accumulating ones (summation order only).
PAIR_FUSE_MAX_TOKENS = 32768


def encode_pair(model, q, d, normalize):
    """(query embeddings, document embeddings) from one tower call, or None when the two-call form applies: other tower
    types (image towers, wrapped models), towers called without a mask-free length list AND without masks of equal
    layout, or sides big enough to fill the chip on their own."""
    if not isinstance(model, BiEncoder) or model.is_vision or model.selector is not None:
        return None
    qi, di = q["input_ids"], d["input_ids"]
    if qi.dim() != 2 or di.dim() != 2 or set(q) != set(d):
        return None
    if not set(q) <= {"input_ids", "attention_mask", "seqlens"}:
        return None   #  a key this function does not forward must not be dropped silently: two-call form
    if qi.shape[1] != di.shape[1] and "attention_mask" not in q and "seqlens" not in q:
        #  widths differ and nothing says where a row ends: the right-padding below would become real tokens of the
        # narrower side (VarlenBatch.from_mask(None) takes every position) -- the two-call form applies
        return None
    if max(qi.shape[0] * qi.shape[1], di.shape[0] * di.shape[1]) > PAIR_FUSE_MAX_TOKENS:
        return None
    S = max(qi.shape[1], di.shape[1])

    def widen(t, fill=0):   # right-pad to the common width (pad positions are masked out / beyond the lengths)
        return t if t.shape[1] == S else torch.nn.functional.pad(t, (0, S - t.shape[1]), value=fill)

    both = {"input_ids": torch.cat([widen(qi), widen(di)], 0)}
    if "attention_mask" in q:
        both["attention_mask"] = torch.cat([widen(q["attention_mask"]), widen(d["attention_mask"])], 0)
    if "seqlens" in q:
        both["seqlens"] = list(np.asarray(q["seqlens"]).reshape(-1)) + list(np.asarray(d["seqlens"]).reshape(-1))
    emb = model(**both, normalize=normalize)["embedding"]
    nq = qi.shape[0]
    return emb[:nq], emb[nq:]


class JsonlTracker:
    """The tracker the trainers hand to clip_loss when `train_args.wandb` is set and the `wandb` package is not importable (it is
    not in this image): the call surface the reference uses -- `.log(dict, step=)` (sc/trainers/base.py:137-139, sc/loss.py:127-130)
    -- appending one JSON object per call to <output_dir>/metrics.jsonl."""

    def __init__(self, path: str):
        import os

        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.path = path
        self.rows = 0

    def log(self, metrics: dict, step=None):
        import json

        with open(self.path, "a") as f:
            f.write(json.dumps({"step": step, **{k: (float(v) if hasattr(v, "__float__") else v) for k, v in metrics.items()}}) + "\n")
        self.rows += 1


class TextTextTrainer:
    def __init__(self, config: Config, dtype=torch.bfloat16, device=None, trunk_config: Optional[NomicBertConfig] = None,
                 total_steps: Optional[int] = None):
        if dtype != torch.bfloat16:
            raise NotImplementedError("the native path computes in bf16 with fp32 master weights (--dtype=bf16)")
        self.accum = _accumulation_steps(config.train_args)
        self.config = config
        from .loss import reset_agreed_free
        reset_agreed_free()   # every rank builds its trainer at the same point: the memory planner's agreed budget starts over
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        self.rank = dist.get_rank() if self.distributed else 0
        torch.manual_seed(config.data_args.seed)
        # MI355X scheduling decisions are config (train_args.gradcache_chunk / gradcache_resident / use_fp8 / exchange),
        # carried explicitly: no process-wide switch survives this trainer
        self.policy = GradCachePolicy.from_train_args(config.train_args)
        set_exchange_mode(config.train_args.exchange or "auto")
        set_exchange_timeout(config.train_args.exchange_timeout_s)
        self.model = self.get_model(config, trunk_config)
        self.total_steps = total_steps or config.train_args.num_train_steps or 10_000
        self.optimizer = self.get_optimizer(config)
        self.scheduler = self.get_scheduler(config, self.optimizer)
        self.step = 0
        # sc/trainers/base.py:42-45,161-184: a tracker exists on rank 0 only, and only when the recipe asks for one
        self.tracker = self.get_trackers(config) if getattr(config.train_args, "wandb", False) else None
        self._need_zero = True      # (gradient accumulation: the buffers are zeroed on the micro-step after an optimizer step)
        # sc/trainers/base.py:387-391: an EMA copy of the weights, updated after every training step
        if self._wants_ema(config):
            from .optimizer import EmaWeights

            self.model["ema"] = EmaWeights([p for g in self.optimizer.param_groups for p in g["params"]], self._ema_decay(config))
            self.model["ema_gettr"] = lambda m: m

    def get_trackers(self, config):
        """sc/trainers/base.py:161-184: wandb.init(...) on rank 0 when the package exists; otherwise a JSON-lines file with the
        same `.log(metrics, step=)` surface.  Ranks > 0 get None, as in the reference (their clip_loss logs nothing)."""
        if self.rank != 0:
            return None
        ta = config.train_args
        run_name = ta.wandb_run_name or (ta.output_dir or "run").replace("ckpts/", "")
        try:
            import wandb  # noqa: F401  (absent from this image; a deployment that has it gets the reference's tracker)
        except ImportError:
            import os

            if not ta.output_dir:   # (the fallback used to write ./metrics.jsonl into whatever the working directory was)
                raise ValueError("train_args.wandb without the wandb package writes <output_dir>/metrics.jsonl: set train_args.output_dir")
            return JsonlTracker(os.path.join(ta.output_dir, "metrics.jsonl"))
        # sc/trainers/base.py:161-184: project, entity, run name, the run's group and the whole config as the run's payload
        payload = config.model_dump() if hasattr(config, "model_dump") else (config.dict() if hasattr(config, "dict") else None)
        return wandb.init(project=ta.wandb_project_name, entity=ta.wandb_entity or None, name=run_name,
                          group=getattr(ta, "wandb_group", None) or None, config=payload)

    def log(self, metrics, step=None):   # sc/trainers/base.py:137-139
        if self.rank == 0 and self.tracker is not None:
            self.tracker.log(metrics, step=step)

    def _wants_ema(self, config) -> bool:
        return bool(getattr(config.model_args, "ema", False)) if config.model_args is not None else False

    def _ema_decay(self, config) -> float:
        return float(config.model_args.ema_decay)

    def set_total_steps(self, total_steps: int):
        """The schedule horizon is baked into the LR lambda: derive it from the dataloader BEFORE training starts
        (sc/trainers/base.py:228-243 computes it from len(dataloader)); this rebuilds the scheduler for a new horizon."""
        if self.step != 0:
            raise RuntimeError("set_total_steps after training started")
        self.total_steps = max(1, int(total_steps))
        self.scheduler = self.get_scheduler(self.config, self.optimizer)

    # sc/trainers/text_text.py:139-182
    def get_model(self, config: Config, trunk_config=None) -> Dict[str, torch.nn.Module]:
        ma = config.model_args
        bc = BiEncoderConfig(model_name=ma.model_name or "", pooling=ma.pooling, logit_scale=ma.logit_scale,
                             trainable_logit_scale=ma.trainable_logit_scale, projection_dim=ma.projection_dim,
                             freeze=ma.freeze, hamming=ma.hamming, gradient_checkpointing=ma.gradient_checkpointing,
                             checkpoint_keep_layers=_parse_keep(config.train_args.checkpoint_keep_layers, "train_args.checkpoint_keep_layers"),
                             nomic_encoder=ma.nomic_encoder, seq_len=ma.seq_len,
                             resid_pdrop=ma.resid_pdrop if ma.pretrained else None,   # (the reference overrides a PRETRAINED trunk only)
                             trunk_config=trunk_config)
        model = BiEncoder(bc, device=self.device).train()
        model.overlap_reduce = bool(config.train_args.overlap_grad_reduce)
        _load_initial_weights(model, ma, explicit_arch=trunk_config is not None)
        model.broadcast_parameters(0)  # what DDP's constructor does
        scale = LogitScale(SimpleNamespace(logit_scale=ma.logit_scale, trainable_logit_scale=ma.trainable_logit_scale))
        return {"model": model, "logit_scale": scale.to(self.device)}

    def _sync_logit_scale_grad(self):
        """A trainable LogitScale is DDP-wrapped in the reference (sc/trainers/text_text.py:172-178): average its
        gradient over ranks like every other parameter."""
        ls = self.model["logit_scale"].logit_scale if "logit_scale" in self.model else None
        if ls is not None and ls.grad is not None and self.distributed and self.world > 1:
            dist.all_reduce(ls.grad)
            ls.grad.div_(self.world)

    # sc/optimizer.py:7-47 (decay / no-decay groups, torch AdamW)
    def get_optimizer(self, config: Config):
        ta = config.train_args
        groups = self.model["model"].param_groups(ta.weight_decay)
        if self.model["logit_scale"].logit_scale.requires_grad:
            groups[1]["params"].append(self.model["logit_scale"].logit_scale)
        return FusedAdamW(groups, lr=ta.learning_rate, betas=(ta.adam_beta1, ta.adam_beta2), eps=ta.eps)

    # sc/trainers/base.py:362-385: clip_grad_norm_(max_grad_norm) (skipped when None / <= 0) + optimizer.step(), fused
    def _clip_and_step(self):
        self.optimizer.step(max_grad_norm=self.config.train_args.max_grad_norm)

    # sc/trainers/base.py:228-265 (warmup_steps or warmup_pct is mandatory there too: quirk 22)
    def get_scheduler(self, config: Config, optimizer):
        ta = config.train_args
        if ta.warmup_steps is None and ta.warmup_pct is None:
            raise ValueError("warmup_steps or warmup_pct must be set")
        warm = ta.warmup_steps if ta.warmup_steps is not None else int(ta.warmup_pct * self.total_steps)
        return torch.optim.lr_scheduler.LambdaLR(optimizer, _lr_lambda(ta.schedule_type, warm, self.total_steps))

    def _inputs(self, batch, prefix):
        out = {"input_ids": batch[f"{prefix}_input_ids"].to(self.device, non_blocking=True)}
        m = batch.get(f"{prefix}_attention_mask")
        if m is not None:
            out["attention_mask"] = m.to(self.device, non_blocking=True)
        if f"{prefix}_seqlens" in batch:
            out["seqlens"] = batch[f"{prefix}_seqlens"]
        return out

    # sc/trainers/text_text.py:276-322 (grad cache) / :324-378 (direct)
    def forward_step(self, batch) -> torch.Tensor:
        ta = self.config.train_args
        model, scale = self.model["model"], self.model["logit_scale"]
        if "negative_input_ids" in batch:
            raise ValueError("negatives must be folded into document_* (sc/trainers/text_text.py:346-347)")
        q, d = self._inputs(batch, "query"), self._inputs(batch, "document")
        if ta.grad_cache:
            loss = grad_cache_loss(model, q, model, d, ta.chunk_size, scale, policy=self.policy)
            self._sync_logit_scale_grad()
            return loss
        dims = ta.matryoshka_dims
        normalize = dims is None  # sc/trainers/text_text.py:325
        pair = encode_pair(model, q, d, normalize)
        if pair is not None:
            queries, documents = pair
        else:
            queries = model(**q, normalize=normalize)["embedding"]
            documents = model(**d, normalize=normalize)["embedding"]
        all_documents = gather_with_grad(documents)
        # sc/trainers/text_text.py:352-378: the tracker (rank 0, `wandb: true`) receives the in-batch accuracy from clip_loss
        # under the dataset's name (per Matryoshka width: `<dataset>_matryoshka_<dim>`)
        dataset_name = batch.get("dataset_name", "") if isinstance(batch, dict) else ""
        if isinstance(dataset_name, (list, tuple)):
            dataset_name = dataset_name[0] if dataset_name else ""
        tracker = getattr(self, "tracker", None)
        log = dict(tracker=tracker, step=getattr(self, "step", None)) if tracker is not None else {}
        if not dims:
            return clip_loss(queries, all_documents, scale, use_fp8=bool(ta.use_fp8), dataset=dataset_name, **log)
        # Matryoshka (text_text.py:352-369): one InfoNCE per prefix width on re-normalised prefixes, weighted sum.
        # The fused loss kernel reads the (N, dim) prefix views in place (leading dimension 768).
        weights = ta.matryoshka_loss_weights or [1.0] * len(dims)
        loss = 0.0
        for w, dim in zip(weights, dims):
            rq = torch.nn.functional.normalize(queries[:, :dim], dim=-1)
            rd = torch.nn.functional.normalize(all_documents[:, :dim], dim=-1)
            loss = loss + w * clip_loss(rq, rd, scale, use_fp8=bool(ta.use_fp8), dataset=f"{dataset_name}_matryoshka_{dim}", **log)
        return loss

    def backward(self, loss: torch.Tensor):
        if self.config.train_args.grad_cache:
            return  # gradients were accumulated inside grad_cache_loss (text_text.py:292-302)
        # the tower ran twice in this graph (queries, documents): the backward that consumes its last saved forward
        # completes its gradients and starts their reduction block by block
        self.model["model"].arm_overlapped_reduce(when_last_outstanding=True)
        loss.backward()
        self.model["model"].sync_gradients()
        self._sync_logit_scale_grad()

    # sc/trainers/base.py:366-393
    def _zero_grads(self):
        self.model["model"].trunk.zero_grad()
        self.optimizer.zero_grad(set_to_none=False)

    def _after_optimizer_step(self):
        self.model["model"].trunk.sync_shadows()

    def _micro_step(self, batch):
        """One call = one micro-batch (sc/trainers/base.py:366-393).  gradient_accumulation_steps == 1 (every shipped
        contrastive recipe): zero, forward, backward, fused clip + AdamW, scheduler.  > 1: the reference's schedule with its
        quirks kept -- `backward` sums the micro-batches' gradients (no averaging), the clip fires on micro-step
        `step % accum == 0` (on whatever has been accumulated by then), optimizer and scheduler on `(step + 1) % accum == 0` or
        on the run's last micro-step, gradients are zeroed right after the optimizer step.  Under data parallelism every
        micro-step's reduction averages the whole gradient buffer: the part accumulated by earlier micro-steps is already
        identical on every rank, so averaging it again leaves it unchanged -- the sum over micro-steps of the averaged
        gradients, which is what DDP produces."""
        ta = self.config.train_args
        if self.accum == 1:
            self._zero_grads()
            out = self.forward_step(batch)
            self.backward(out)
            self._clip_and_step()
            self.scheduler.step()
            self._after_optimizer_step()
        else:
            if self._need_zero:
                self._zero_grads()
                self._need_zero = False
            out = self.forward_step(batch)
            self.backward(out)
            clip = ta.max_grad_norm is not None and ta.max_grad_norm > 0
            if clip and self.step % self.accum == 0:
                params = [p for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
                torch.nn.utils.clip_grad_norm_(params, ta.max_grad_norm)
            if (self.step + 1) % self.accum == 0 or self.step == self.total_steps - 1:
                self.optimizer.step(max_grad_norm=None)
                self.scheduler.step()
                self._after_optimizer_step()
                self._need_zero = True
        if self.model.get("ema") is not None:   # (every micro-step, as the reference does)
            self.model["ema"].update(self.model["ema_gettr"](self.model["model"]))
        return out

    def training_step(self, batch) -> torch.Tensor:
        loss = self._micro_step(batch)
        self.step += 1
        if self.world > 1:
            check_exchange(sync=True)   # a peer that never signalled stops this step, not the next one
        return loss.detach()

    # ---- checkpoint / resume (sc/trainers/base.py:275-344: <dir>/model, optimizer.pt, scheduler.pt,
    #      random_states_<rank>.pt; the streaming loader's per-rank offsets are the data side's business) ------------
    def _towers(self):
        m = self.model["model"]
        return {"model": m} if hasattr(m, "trunk") else {"text": m.text, "vision": m.vision}

    def _logit_scale_param(self):
        if "logit_scale" in self.model:
            return self.model["logit_scale"].logit_scale
        m = self.model["model"]
        return m.logit_scale.logit_scale if hasattr(m, "logit_scale") else None

    def save_state(self, output_dir: str):
        import os
        import random

        os.makedirs(output_dir, exist_ok=True)
        if self.rank == 0:
            for name, tower in self._towers().items():
                tower.save_pretrained(os.path.join(output_dir, name))
            ls = self._logit_scale_param()
            if ls is not None and ls.requires_grad:  # sc/trainers/text_text.py:247-255: saved next to the towers
                torch.save({"logit_scale": ls.detach().cpu()}, os.path.join(output_dir, "logit_scale.pt"))
            torch.save(self.optimizer.state_dict(), os.path.join(output_dir, "optimizer.pt"))
            # reference layout (sc/trainers/base.py:275-344): scheduler.pt IS the scheduler's state_dict; the step counter
            # travels in its own file
            torch.save(self.scheduler.state_dict(), os.path.join(output_dir, "scheduler.pt"))
            torch.save({"step": self.step}, os.path.join(output_dir, "trainer_state.pt"))
            if self.model.get("ema") is not None:
                torch.save(self.model["ema"].state_dict(), os.path.join(output_dir, "ema.pt"))
        torch.save({"torch": torch.get_rng_state(), "numpy": np.random.get_state(), "random": random.getstate(),
                    "cuda": torch.cuda.get_rng_state_all()}, os.path.join(output_dir, f"random_states_{self.rank}.pt"))
        if self.world > 1:   # rank 0 wrote ~2 GB: nobody runs ahead into the next step's exchange meanwhile
            import torch.distributed as dist

            dist.barrier()

    def load_state(self, input_dir: str):
        import os
        import random

        for name, tower in self._towers().items():
            tower.load_pretrained(os.path.join(input_dir, name))
        self.optimizer.load_state_dict(torch.load(os.path.join(input_dir, "optimizer.pt"), map_location=self.device))
        ls = self._logit_scale_param()
        ls_path = os.path.join(input_dir, "logit_scale.pt")
        if ls is not None and os.path.exists(ls_path):
            with torch.no_grad():
                ls.copy_(torch.load(ls_path)["logit_scale"].to(ls.device))
        sch = torch.load(os.path.join(input_dir, "scheduler.pt"))
        if "scheduler" in sch and "step" in sch:  # round-1 layout
            self.scheduler.load_state_dict(sch["scheduler"])
            self.step = int(sch["step"])
        else:
            self.scheduler.load_state_dict(sch)
            st = os.path.join(input_dir, "trainer_state.pt")
            self.step = int(torch.load(st)["step"]) if os.path.exists(st) else int(sch.get("last_epoch", 0))
        if self.model.get("ema") is not None and os.path.exists(os.path.join(input_dir, "ema.pt")):
            self.model["ema"].load_state_dict(torch.load(os.path.join(input_dir, "ema.pt")))
        rs = torch.load(os.path.join(input_dir, f"random_states_{self.rank}.pt"), weights_only=False)
        torch.set_rng_state(rs["torch"])
        np.random.set_state(rs["numpy"])
        random.setstate(rs["random"])
        torch.cuda.set_rng_state_all(rs["cuda"])

    def train(self, batches: Iterable[dict], max_steps: Optional[int] = None, log_every: int = 0):
        losses = []
        for i, batch in enumerate(batches):
            if max_steps is not None and i >= max_steps:
                break
            loss = self.training_step(batch)
            losses.append(loss)
            if bool(getattr(self.config.train_args, "wandb", False)):   # sc/trainers/base.py:485-502 (loss every step; lr every log_lr_every steps)
                # the reference gathers the loss from every rank and logs the mean (rank 0's own value made multi-GPU curves
                # incomparable with reference runs); every rank takes part in the reduction, rank 0 alone has a tracker
                lv = loss.detach().float()
                if self.world > 1:
                    import torch.distributed as dist

                    lv = lv.clone()
                    dist.all_reduce(lv)
                    lv /= self.world
                self.log({"loss": float(lv)}, step=self.step - 1)
                every = int(getattr(self.config.train_args, "log_lr_every", 0) or 0)
                if every and i > 0 and i % every == 0:
                    self.log({"lr": self.scheduler.get_last_lr()[0]}, step=self.step - 1)
            if log_every and (i + 1) % log_every == 0 and self.rank == 0:
                print(f"step {self.step} loss {float(loss):.4f} lr {self.scheduler.get_last_lr()[0]:.3e}", flush=True)
        return losses


class ImageTextTrainer(TextTextTrainer):
    """CLIP-style / LiT image-text contrastive training (host-side mirror of sc/trainers/image_text.py:24-196) on the
    native towers: `DualEncoder(text BiEncoder, vision BiEncoder over a ViTEngine, LogitScale)`.  A tower with
    `freeze: true` (LiT's image tower) runs forward-only and is left out of the optimizer and the gradient all-reduce.
    Batches carry {"text": {...BiEncoder kwargs...}, "vision": {"input_ids": pixels (B,3,H,W)}} as in the reference."""

    def __init__(self, config: Config, dtype=torch.bfloat16, device=None, text_trunk_config=None,
                 vision_trunk_config=None, total_steps: Optional[int] = None):
        self._trunks = (text_trunk_config, vision_trunk_config)
        super().__init__(config, dtype=dtype, device=device, total_steps=total_steps)

    # sc/trainers/image_text.py:53-68 + sc/models/dual_encoder/modeling_dual_encoder.py:10-24
    def get_model(self, config: Config, trunk_config=None) -> Dict[str, torch.nn.Module]:
        if config.text_model_args is None or config.vision_model_args is None:
            raise ValueError("image_text needs text_model_args and vision_model_args")
        if config.tower_model_args is not None:
            raise NotImplementedError("three-tower image-text training")
        towers = []
        for ma, trunk in zip((config.text_model_args, config.vision_model_args), self._trunks):
            bc = BiEncoderConfig(model_name=ma.model_name or "", pooling=ma.pooling, logit_scale=ma.logit_scale,
                                 trainable_logit_scale=ma.trainable_logit_scale, projection_dim=ma.projection_dim,
                                 freeze=ma.freeze, hamming=ma.hamming, gradient_checkpointing=ma.gradient_checkpointing,
                                 checkpoint_keep_layers=_parse_keep(config.train_args.checkpoint_keep_layers, "train_args.checkpoint_keep_layers"),
                                 nomic_encoder=ma.nomic_encoder,
                                 resid_pdrop=ma.resid_pdrop if ma.pretrained else None,
                                 patch_dropout=float(ma.patch_dropout or 0.0),
                                 seq_len=ma.seq_len, trunk_config=trunk)
            tower = BiEncoder(bc, device=self.device).train()
            tower.overlap_reduce = bool(config.train_args.overlap_grad_reduce)
            _load_initial_weights(tower, ma, explicit_arch=trunk is not None)
            tower.broadcast_parameters(0)
            towers.append(tower)
        va = config.vision_model_args  # the reference takes the logit scale from the image tower's args
        scale = LogitScale(SimpleNamespace(logit_scale=va.logit_scale, trainable_logit_scale=va.trainable_logit_scale))
        model = DualEncoder(towers[0], towers[1], scale.to(self.device),
                            precomputed_text=bool(config.text_model_args.precomputed),
                            use_fp8=bool(config.train_args.use_fp8)).train()
        return {"model": model}

    def _trainable_towers(self):
        m = self.model["model"]
        return [t for t in (m.text, m.vision) if not t.frozen_trunk]

    def get_optimizer(self, config: Config):
        ta = config.train_args
        groups = [{"params": [], "weight_decay": ta.weight_decay}, {"params": [], "weight_decay": 0.0}]
        for t in self._trainable_towers():
            g = t.param_groups(ta.weight_decay)
            groups[0]["params"] += g[0]["params"]
            groups[1]["params"] += g[1]["params"]
        ls = self.model["model"].logit_scale.logit_scale
        if ls.requires_grad:
            groups[1]["params"].append(ls)
        return FusedAdamW(groups, lr=ta.learning_rate, betas=(ta.adam_beta1, ta.adam_beta2), eps=ta.eps)

    # sc/trainers/image_text.py:154-178
    def forward_step(self, batch):
        if self.config.train_args.grad_cache:
            raise NotImplementedError("Grad cache not supported for three towers")  # the reference's own refusal
        text = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch["text"].items()}
        vision = {k: v.to(self.device, non_blocking=True) for k, v in batch["vision"].items()}
        return self.model["model"](text, vision)

    def backward(self, loss):
        for t in self._trainable_towers():
            t.arm_overlapped_reduce(when_last_outstanding=True)
        loss["loss"].backward()
        for t in self._trainable_towers():
            t.sync_gradients()
        ls = self.model["model"].logit_scale.logit_scale
        if ls.grad is not None and self.distributed and self.world > 1:
            dist.all_reduce(ls.grad)
            ls.grad.div_(self.world)

    def _zero_grads(self):
        for t in self._trainable_towers():
            t.trunk.zero_grad()
        self.optimizer.zero_grad(set_to_none=False)

    def _after_optimizer_step(self):
        for t in self._trainable_towers():
            t.trunk.sync_shadows()
        ta = self.config.train_args
        ls = self.model["model"].logit_scale.logit_scale
        if ta.clamp_logits and ls.requires_grad:
            with torch.no_grad():
                ls.clamp_(0, float(np.log(ta.logit_max)))

    def _wants_ema(self, config) -> bool:
        return any(bool(getattr(ma, "ema", False)) for ma in (config.text_model_args, config.vision_model_args) if ma is not None)

    def _ema_decay(self, config) -> float:
        return float((config.vision_model_args or config.text_model_args).ema_decay)

    # sc/trainers/base.py:366-393 + image_text.py:180-196 (logit clamp)
    def training_step(self, batch) -> torch.Tensor:
        out = self._micro_step(batch)
        self.step += 1
        if self.world > 1:
            check_exchange(sync=True)
        return out["loss"].detach()


# sc/trainers/__init__.py:9-17 (the contrastive entries + MLM pretraining; glue / distill trainers are outside the hot path)
def _mlm_trainer(*a, **k):
    from .mlm import MLMTrainer  # imported lazily: mlm.py imports this module's schedule helper

    return MLMTrainer(*a, **k)


TRAINER_REGISTRY = {"encoder": TextTextTrainer, "image_text": ImageTextTrainer, "locked_text": ImageTextTrainer,
                    "mlm": _mlm_trainer}


def synthetic_batches(n_steps: int, per_rank_batch: int, seq_len: int, vocab: int = 30522, seed: int = 1234,
                      rank: int = 0, ragged: bool = False):
    """Synthetic (query, document) batches with the streaming loader's keys (SURVEY.md §8d recipe)."""
    g = torch.Generator().manual_seed(seed + rank)
    for _ in range(n_steps):
        b = {}
        for side in ("query", "document"):
            ids = torch.randint(min(1000, vocab // 2), vocab, (per_rank_batch, seq_len), generator=g)
            ids[:, 0] = 101
            lens = torch.randint(seq_len // 2, seq_len + 1, (per_rank_batch,), generator=g) if ragged else \
                torch.full((per_rank_batch,), seq_len)
            mask = (torch.arange(seq_len)[None] < lens[:, None]).long()
            b[f"{side}_input_ids"] = ids * mask
            b[f"{side}_attention_mask"] = mask
            b[f"{side}_seqlens"] = lens.numpy()
        b["dataset_name"] = "synthetic"
        yield b
