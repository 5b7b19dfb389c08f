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


# ---- selective checkpointing (round 3): the top k blocks keep their activations, only the rest are recomputed -------------
@pytest.mark.parametrize("arch", ["nomic", "bert"])
def test_selective_checkpointing_is_bit_identical_for_every_keep_count(arch):
    """CxChunkBuffers.ckpt_keep = 0 .. L: same embeddings, same Linear / LayerNorm gradients bit for bit as the engine
    without checkpointing (keep = L is that engine's schedule on a checkpointing arena), arena size strictly in between."""
    L = 4
    cfg = (NomicBertConfig.nomic_bert_2048(vocab_size=2048, n_layer=L) if arch == "nomic"
           else NomicBertConfig.bert_base_uncased(vocab_size=2048, n_layer=L))
    ns = SimpleNamespace(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    sd = encoder_ref.random_state_dict(ns, 12)
    ids, lens, g = _ragged(5, 200, 2048, 6)
    vb = VarlenBatch.from_lengths(ids, lens)
    probe = torch.randn(5, cfg.n_embd, generator=g).to(DEV)

    def run(ck, keep):
        eng = NomicBertEngine(cfg, device=DEV)
        eng.load_reference_state_dict(sd)
        eng.train()
        eng.gradient_checkpointing_enable(ck, keep_layers=keep)
        emb, arena = eng.forward_chunk(vb, True)
        assert arena.checkpoint == ck and arena.keep_layers == (keep if ck else 0)
        nbytes = arena.nbytes()
        eng.zero_grad()
        eng.backward_chunk(vb, arena, probe)
        torch.cuda.synchronize()
        return emb.clone(), eng.reference_grad_dict(), nbytes

    emb0, g0, full_bytes = run(False, 0)
    sizes = []
    for keep in range(L + 1):
        emb, gk, nbytes = run(True, keep)
        sizes.append(nbytes)
        assert torch.equal(emb0, emb), keep
        for name, a in g0.items():
            b = gk[name]
            if ".layers." in name and not name.endswith(".bias"):
                assert torch.equal(a, b), (keep, name, float((a - b).abs().max()))
            else:
                assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-30, (keep, name)
    assert all(x < y for x, y in zip(sizes, sizes[1:])), sizes
    report("checkpoint_selective", arch=arch, arena_bytes_by_keep=sizes, arena_bytes_full=full_bytes)


def test_selective_checkpointing_vit_tower_is_bit_identical():
    from contrastors_amd.vit import ViTConfig, ViTEngine

    cfg = ViTConfig(n_embd=256, n_layer=3, n_head=4, n_inner=1024, img_size=64, patch_size=16)
    g = torch.Generator().manual_seed(9)
    pixels = torch.randn(6, 3, 64, 64, generator=g).to(DEV)
    probe = torch.randn(6, cfg.n_embd, generator=g).to(DEV)
    res = {}
    for key, (ck, keep) in {"plain": (False, 0), "k1": (True, 1), "k2": (True, 2), "k3": (True, 3)}.items():
        eng = ViTEngine(cfg, device=DEV, pooling="cls", seed=5)
        eng.train()
        eng.gradient_checkpointing_enable(ck, keep_layers=keep)
        emb, arena = eng.forward_chunk(pixels, True)
        assert arena.keep_layers == keep
        eng.zero_grad()
        eng.backward_chunk(pixels, arena, probe)
        torch.cuda.synchronize()
        res[key] = (emb.clone(), eng.flat_grad.clone())
    for key in ("k1", "k2", "k3"):
        assert torch.equal(res["plain"][0], res[key][0]), key
        assert torch.equal(res["plain"][1], res[key][1]), (key, float((res["plain"][1] - res[key][1]).abs().max()))


def test_auto_keep_measures_the_first_use_then_keeps_what_fits():
    """checkpoint_keep_layers = 'auto': the first use of an arena takes the recipe literally (every block recomputed); once
    its backward has run the step's peak HBM is known and the arena's successor keeps what the headroom pays for -- all
    four blocks of this small problem -- with the same embeddings and gradients; the per-device ledger is settled when
    the arena dies; a suspended engine (resident GradCache lines up one arena per chunk) stays literal; CX_CHECKPOINT_KEEP
    overrides the config."""
    import gc
    import os

    from contrastors_amd import nomic_bert as nb

    cfg = NomicBertConfig.nomic_bert_2048(vocab_size=2048, n_layer=4)
    eng = NomicBertEngine(cfg, device=DEV, seed=1)
    eng.train()
    eng.gradient_checkpointing_enable(True, keep_layers="auto")
    ids, lens, g = _ragged(4, 128, 2048, 2)
    vb = VarlenBatch.from_lengths(ids, lens)
    probe = torch.randn(4, cfg.n_embd, generator=g).to(DEV)
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    ledger0 = nb._hbm_grant(eng.device_, 0)
    res = []
    for step in range(3):
        emb, arena = eng.forward_chunk(vb, True)
        assert arena.keep_layers == (0 if step == 0 else 4), (step, arena.keep_layers)
        assert arena.probation == (step == 0)
        eng.zero_grad()
        eng.backward_chunk(vb, arena, probe)
        torch.cuda.synchronize()
        res.append((emb.clone(), eng.flat_grad.clone()))
    del arena
    assert nb._hbm_grant(eng.device_, 0) > ledger0 and eng._keep_plan[vb.T if vb.T % 128 == 0 else (vb.T + 127) // 128 * 128] == 4
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[1][0], res[2][0])
    lin = slice(eng._lin_begin, eng._lin_end)   # Linear weights: bit-identical; the rest carries fp32-atomics order noise
    assert torch.equal(res[0][1][lin], res[1][1][lin]) and torch.equal(res[1][1][lin], res[2][1][lin])
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * float(res[0][1].abs().max())
    eng._arena_free.clear()
    gc.collect()
    assert nb._hbm_grant(eng.device_, 0) == ledger0      # the arena returned its grant
    eng._keep_plan.clear()
    with eng.selective_checkpointing_suspended():
        for _ in range(2):
            _, arena = eng.forward_chunk(vb, True)
            assert arena.keep_layers == 0 and not arena.probation
            eng.backward_chunk(vb, arena, probe)
    eng._arena_free.clear()
    os.environ["CX_CHECKPOINT_KEEP"] = "1"
    try:
        _, arena = eng.forward_chunk(vb, True)
        assert arena.keep_layers == 1 and not arena.probation
        eng.abandon_arena(arena)
    finally:
        del os.environ["CX_CHECKPOINT_KEEP"]


@pytest.mark.parametrize("attn_pdrop", [0.0, 0.1])
def test_checkpointing_under_dropout_regenerates_the_same_masks(attn_pdrop):
    """resid_pdrop / embd_pdrop (/ attn_pdrop) > 0: a recomputed block must draw the masks its first forward drew -- they are a
    pure function of the chunk's Philox (seed, offset) in the arena, the dropout site and the indices -- so literal and
    selective checkpointing give the embeddings and gradients of the non-checkpointed engine under the same generator state
    (max_seqlen > 128 with attention dropout: the streaming kernels are the ones that carry the mask code)."""
    L, S = 4, 160 if attn_pdrop > 0 else 96
    cfg = NomicBertConfig.nomic_bert_2048(vocab_size=1024, n_layer=L, resid_pdrop=0.1, embd_pdrop=0.1, attn_pdrop=attn_pdrop)
    ids, lens, g = _ragged(6, S, 1024, 21)
    vb = VarlenBatch.from_lengths(ids, lens)
    probe = torch.randn(6, cfg.n_embd, generator=g).to(DEV)
    res = {}
    p0 = None
    for key, (ck, keep) in {"plain": (False, 0), "literal": (True, 0), "keep2": (True, 2)}.items():
        eng = NomicBertEngine(cfg, device=DEV, seed=3)
        if p0 is None:
            p0 = eng.flat_param.clone()
        eng.flat_param.copy_(p0)
        eng.sync_shadows()
        eng.train()
        eng.gradient_checkpointing_enable(ck, keep_layers=keep)
        torch.manual_seed(123)   # the device generator the engine draws (seed, offset) from
        emb, arena = eng.forward_chunk(vb, True)
        assert arena.desc.drop_active == 1 and arena.keep_layers == keep
        eng.zero_grad()
        eng.backward_chunk(vb, arena, probe)
        torch.cuda.synchronize()
        res[key] = (emb.clone(), {k: v.clone() for k, v in eng.reference_grad_dict().items()})   # (views of flat_grad)
    torch.manual_seed(124)
    emb_other, arena = eng.forward_chunk(vb, True)
    eng.abandon_arena(arena)
    assert not torch.equal(emb_other, res["plain"][0]), "another generator state must give other masks"
    for key in ("literal", "keep2"):
        assert torch.equal(res["plain"][0], res[key][0]), key
        for name, a in res["plain"][1].items():
            b = res[key][1][name]
            if ".layers." in name and not name.endswith(".bias"):
                assert torch.equal(a, b), (key, name, float((a - b).abs().max()))
            else:
                assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-30, (key, name)


def test_ragged_batches_ratchet_one_arena_up_and_idle_arenas_are_given_back():
    """Token counts change every step with ragged batches.  An arena serves every batch up to its capacity; a new record
    builds a bigger one (one literal step, then its kept blocks); the arena left behind is given back once it has sat idle
    for ARENA_IDLE_USES saving forwards, its ledger grant with it -- nothing accumulates."""
    import gc

    from contrastors_amd import nomic_bert as nb

    cfg = NomicBertConfig.nomic_bert_2048(vocab_size=1024, n_layer=2)
    eng = NomicBertEngine(cfg, device=DEV, seed=2)
    eng.train()
    eng.gradient_checkpointing_enable(True, keep_layers="auto")
    eng.ARENA_IDLE_USES = 4
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    ledger0 = nb._hbm_grant(eng.device_, 0)
    g = torch.Generator().manual_seed(5)

    def step(total_tokens):
        B = 80   # (a fixed batch of 80 sequences whose lengths change: what a loader hands over)
        lens = [total_tokens // B + (1 if i < total_tokens % B else 0) for i in range(B)]
        ids = torch.randint(3, 1024, (B, 128), generator=g).to(DEV)
        vb = VarlenBatch.from_lengths(ids, lens)
        emb, arena = eng.forward_chunk(vb, True)
        info = (arena.T_cap, arena.keep_layers, arena.probation)
        eng.zero_grad()
        eng.backward_chunk(vb, arena, torch.randn(len(lens), cfg.n_embd, generator=g).to(DEV))
        del arena
        return info

    a = step(3000)                       # first batch: literal, measured
    assert a[1:] == (0, True)
    b = step(2500)                       # a smaller batch: the rebuilt arena (every block kept) serves it
    assert b[0] == a[0] and b[1:] == (2, False)
    c = step(5000)                       # a record: a bigger arena, literal once ...
    assert c[0] > a[0] and c[1:] == (0, True)
    d = step(4000)                       # ... then rebuilt with its kept blocks; the 3000-token arena idles in the free list
    assert d[0] == c[0] and d[1:] == (2, False)
    assert sorted(x.T_cap for x in eng._arena_free) == [a[0], c[0]]
    for _ in range(3):
        assert step(4500)[0] == c[0]
    assert step(9000)[1:] == (0, True)   # the next record first gives the idle 3000-token arena back
    torch.cuda.synchronize()
    gc.collect()
    assert a[0] not in [x.T_cap for x in eng._arena_free] and a[0] not in eng._keep_plan
    held = sum(x.granted for x in eng._arena_free) + sum(eng._keep_granted.values())
    assert nb._hbm_grant(eng.device_, 0) - ledger0 == held, (nb._hbm_grant(eng.device_, 0), ledger0, held)


def test_out_of_memory_on_a_record_batch_gives_the_idle_arenas_back_and_retries(monkeypatch):
    """A batch larger than any arena so far, while the outgrown arena (with its kept blocks) still sits in the free list:
    the first allocation attempt may not fit.  Every idle arena is given back and the allocation is tried once more."""
    from contrastors_amd import nomic_bert as nb

    cfg = NomicBertConfig.nomic_bert_2048(vocab_size=1024, n_layer=2)
    eng = NomicBertEngine(cfg, device=DEV, seed=2)
    eng.train()
    eng.gradient_checkpointing_enable(True, keep_layers=0)
    g = torch.Generator().manual_seed(6)

    def batch(n):
        ids = torch.randint(3, 1024, (n, 128), generator=g).to(DEV)
        return VarlenBatch.from_lengths(ids, [128] * n), torch.randn(n, cfg.n_embd, generator=g).to(DEV)

    vb, probe = batch(8)
    _, arena = eng.forward_chunk(vb, True)
    eng.backward_chunk(vb, arena, probe)
    del arena
    assert len(eng._arena_free) == 1
    real, calls = nb._ChunkArena, []

    class Flaky(real):
        def __init__(self, *a, **k):
            calls.append(1)
            if len(calls) == 1:
                raise torch.OutOfMemoryError("simulated")
            super().__init__(*a, **k)

    monkeypatch.setattr(nb, "_ChunkArena", Flaky)
    vb2, probe2 = batch(16)
    emb, arena = eng.forward_chunk(vb2, True)
    assert len(calls) == 2 and eng._arena_free == []
    eng.zero_grad()
    eng.backward_chunk(vb2, arena, probe2)
    torch.cuda.synchronize()
    assert torch.isfinite(emb).all() and torch.isfinite(eng.flat_grad).all() and float(eng.flat_grad.abs().max()) > 0
    # with nothing idle to give back the error is the caller's
    calls.clear()
    eng._arena_free = []
    with pytest.raises(torch.OutOfMemoryError):
        eng.forward_chunk(batch(32)[0], True)
