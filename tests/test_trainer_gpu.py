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


def _trainer(grad_cache: bool):
    cfg = Config(train_args=TrainArgs(learning_rate=1e-3, weight_decay=0.01, warmup_steps=1, grad_cache=grad_cache,
                                      chunk_size=4, schedule_type="linear", max_grad_norm=1.0, clamp_logits=False),
                 data_args=DataArgs(batch_size=16, seed=7),
                 model_args=ModelArgs(logit_scale=20.0, pooling="mean", model_name="tiny"))
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
