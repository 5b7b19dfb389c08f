"""Activation checkpointing inside the native engine (BiEncoderConfig.gradient_checkpointing;
sc/models/encoder/modeling_nomic_bert.py:339-365, sc/models/vit/vit.py:200-231, enabled by
sc/models/biencoder/modeling_biencoder.py:261-262): a saving forward keeps one (T, d) tensor per block, backward
recomputes each block from it.  Every kernel on the path is deterministic, so embeddings AND gradients must be
bit-identical to the non-checkpointed engine (the word-embedding scatter uses fp32 atomics: atomics noise only)."""
from types import SimpleNamespace

import pytest
import torch

from contrastors_amd.nomic_bert import NomicBertConfig, NomicBertEngine, VarlenBatch
from oracle import encoder_ref
from tests.gpu_util import report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ragged(B, S, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(S // 3, S + 1, (B,), generator=g)
    lens[0] = S
    ids = torch.randint(3, vocab, (B, S), generator=g)
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    return (ids * mask).to(DEV), lens.numpy(), g


@pytest.mark.parametrize("arch", ["nomic", "bert"])
def test_checkpointed_text_trunk_is_bit_identical(arch):
    cfg = (NomicBertConfig.nomic_bert_2048(vocab_size=2048, n_layer=4) if arch == "nomic"
           else NomicBertConfig.bert_base_uncased(vocab_size=2048, n_layer=4))
    ns = SimpleNamespace(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    sd = encoder_ref.random_state_dict(ns, 11)
    ids, lens, g = _ragged(5, 200, 2048, 4)   # T not a multiple of 64: the wgrad pad rows are exercised too
    vb = VarlenBatch.from_lengths(ids, lens)
    probe = torch.randn(5, cfg.n_embd, generator=g).to(DEV)
    out = {}
    for ck in (False, True):
        eng = NomicBertEngine(cfg, device=DEV)
        eng.load_reference_state_dict(sd)
        eng.train()
        eng.gradient_checkpointing_enable(ck)
        emb, arena = eng.forward_chunk(vb, True)
        assert arena.checkpoint == ck
        nbytes = arena.nbytes()
        eng.zero_grad()
        eng.backward_chunk(vb, arena, probe)
        torch.cuda.synchronize()
        out[ck] = (emb.clone(), eng.flat_grad.clone(), nbytes, eng)
    assert torch.equal(out[False][0], out[True][0])
    ga, gb = out[False][3].reference_grad_dict(), out[True][3].reference_grad_dict()
    for name, a in ga.items():
        b = gb[name]
        if ".layers." in name and not name.endswith(".bias"):
            assert torch.equal(a, b), (name, float((a - b).abs().max()))
        else:  # embedding tables, emb_ln and Linear biases are reduced with fp32 atomics: summation-order noise only
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-30, name
    ratio = out[True][2] / out[False][2]
    report("checkpoint_text", arch=arch, arena_bytes_full=out[False][2], arena_bytes_ckpt=out[True][2], ratio=ratio)
    assert ratio < 0.8   # 4 layers, backward scratch included (12 layers at the bench chunk: 0.25)


def test_checkpointed_vit_tower_is_bit_identical():
    from contrastors_amd.vit import ViTConfig, ViTEngine

    cfg = ViTConfig(n_embd=256, n_layer=3, n_head=4, n_inner=1024, img_size=64, patch_size=16)
    g = torch.Generator().manual_seed(8)
    pixels = torch.randn(6, 3, 64, 64, generator=g).to(DEV)
    probe = torch.randn(6, cfg.n_embd, generator=g).to(DEV)
    res = {}
    for ck in (False, True):
        eng = ViTEngine(cfg, device=DEV, pooling="cls", seed=5)
        eng.train()
        eng.gradient_checkpointing_enable(ck)
        emb, arena = eng.forward_chunk(pixels, True)
        assert arena.checkpoint == ck
        eng.zero_grad()
        eng.backward_chunk(pixels, arena, probe)
        torch.cuda.synchronize()
        res[ck] = (emb.clone(), eng.flat_grad.clone())
    assert torch.equal(res[False][0], res[True][0])
    assert torch.equal(res[False][1], res[True][1]), float((res[False][1] - res[True][1]).abs().max())
