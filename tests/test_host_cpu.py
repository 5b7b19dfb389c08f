"""Host logic that needs no GPU: the reference's own recipe YAML parses into our Config, validators, scheduler shape,
varlen index construction, label arithmetic, synthetic batch contract."""
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from contrastors_amd.config import Config, TrainArgs, read_config
from contrastors_amd.loss import make_labels
from contrastors_amd.nomic_bert import VarlenBatch
from contrastors_amd.trainers import _lr_lambda, synthetic_batches

REF_YAML = Path("/root/reference/src/contrastors/configs/train/contrastive_pretrain.yaml")


@pytest.mark.skipif(not REF_YAML.exists(), reason="reference tree only exists in the build container")
def test_reference_recipe_yaml_loads_unchanged():
    cfg = read_config(str(REF_YAML))
    assert cfg.train_args.grad_cache is True and cfg.train_args.chunk_size == 64
    assert cfg.train_args.learning_rate == 2e-4 and cfg.train_args.schedule_type == "cosine"
    assert cfg.model_args.logit_scale == 50 and cfg.model_args.pooling == "mean"
    assert cfg.data_args.batch_size == 16384


def test_validators_match_reference():
    with pytest.raises(ValueError):  # sc/config.py:70-77
        TrainArgs(grad_cache=True, matryoshka_dims=[768, 512])
    with pytest.raises(ValueError):  # sc/config.py:57-68
        TrainArgs(eval_strategy="steps")
    with pytest.raises(ValueError):
        Config(train_args=TrainArgs(), model_args={"model_type": "nope"})


def test_scheduler_shapes():
    f = _lr_lambda("cosine", 700, 10000)
    assert f(0) == 0.0 and f(1) == pytest.approx(1 / 700) and f(700) == pytest.approx(1.0) and f(10000) == pytest.approx(0.0, abs=1e-9)
    g = _lr_lambda("linear", 10, 110)
    assert g(60) == pytest.approx(0.5)
    with pytest.raises(ValueError):
        _lr_lambda("polynomial", 1, 10)


@pytest.mark.parametrize("name", ["cosine", "linear", "constant", "constant_with_warmup", "inverse_sqrt"])
@pytest.mark.parametrize("warmup,total", [(0, 12), (4, 12), (700, 3000)])
def test_lr_schedule_is_the_one_the_reference_builds(name, warmup, total):
    """sc/trainers/base.py:258-263 calls transformers.get_scheduler(name=schedule_type, num_warmup_steps, num_training_steps
    (None for inverse_sqrt)): our LambdaLR must produce the same learning rate at every step, past the horizon included."""
    transformers = pytest.importorskip("transformers")
    ref_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=3e-4)
    ref = transformers.get_scheduler(name=name, optimizer=ref_opt, num_warmup_steps=warmup,
                                     num_training_steps=(total if name != "inverse_sqrt" else None))
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=3e-4)
    ours = torch.optim.lr_scheduler.LambdaLR(opt, _lr_lambda(name, warmup, total))
    for step in range(min(total + 5, 900)):
        assert opt.param_groups[0]["lr"] == pytest.approx(ref_opt.param_groups[0]["lr"], rel=1e-12, abs=1e-18), (name, step)
        opt.step(); ref_opt.step()
        ours.step(); ref.step()


def test_varlen_batch_from_lengths_matches_mask_path():
    ids = torch.arange(4 * 6).view(4, 6)
    lens = [6, 2, 5, 1]
    mask = (torch.arange(6)[None] < torch.tensor(lens)[:, None]).long()
    a = VarlenBatch.from_lengths(ids, lens)
    b = VarlenBatch.from_mask(ids, mask)
    assert a.T == b.T == 14 and a.max_seqlen == b.max_seqlen == 6
    assert torch.equal(a.indices, b.indices) and torch.equal(a.cu_seqlens, b.cu_seqlens)
    assert a.indices.dtype == torch.int32 and a.cu_seqlens.tolist() == [0, 6, 8, 13, 14]
    full = VarlenBatch.from_lengths(ids, [6] * 4)
    assert full.T == 24 and full.indices.tolist() == list(range(24))


def test_labels_int64_with_negatives():
    lab = make_labels(4, 4 * 8 * 2, 1, 2, "cpu")  # 1 positive + 7 negatives per query, rank 1 of 2
    assert lab.dtype == torch.int64
    np.testing.assert_array_equal(lab.numpy(), (np.arange(4) + 4) * 8)


def test_synthetic_batches_follow_loader_contract():
    b = next(synthetic_batches(1, 4, 16, ragged=True))
    for k in ("query_input_ids", "query_attention_mask", "document_input_ids", "document_attention_mask", "dataset_name"):
        assert k in b
    assert b["query_input_ids"].dtype == torch.int64 and b["query_input_ids"].shape == (4, 16)
    assert (b["query_input_ids"][b["query_attention_mask"] == 0] == 0).all()  # right padding with pad id 0


@pytest.mark.skipif(not REF_YAML.exists(), reason="reference tree only exists in the build container")
def test_reference_mlm_recipe_yaml_loads_unchanged():
    """configs/train/mlm.yaml: the architecture overrides the MLM trainer reads (sc/trainers/mlm.py:20-40) survive."""
    cfg = read_config(str(REF_YAML.parent / "mlm.yaml"))
    ma, ta = cfg.model_args, cfg.train_args
    assert ma.model_type == "mlm" and ma.seq_len == 2048 and ma.rotary_emb_base == 500000
    assert ma.activation_function == "swiglu" and ma.rotary_emb_fraction == 1.0 and ma.pad_vocab_to_multiple_of == 64
    assert ma.qkv_proj_bias is False and ma.mlp_fc1_bias is False and ma.mlp_fc2_bias is False
    assert ta.gradient_accumulation_steps == 4 and ta.max_grad_norm == 0.0 and ta.warmup_pct == 0.06
    assert cfg.data_args.mlm_prob == 0.30
    from contrastors_amd.trainers import TRAINER_REGISTRY

    assert set(TRAINER_REGISTRY) >= {"encoder", "image_text", "locked_text", "mlm"}


def test_pretrained_without_local_weights_raises():
    """`pretrained: true` (the reference default, sc/config.py:161) must not silently train from scratch."""
    from contrastors_amd.config import ModelArgs
    from contrastors_amd.trainers import _load_initial_weights

    ma = ModelArgs(model_name="nomic-ai/nomic-bert-2048")
    assert ma.pretrained is True and ma.checkpoint is None
    with pytest.raises(FileNotFoundError):
        _load_initial_weights(object(), ma, explicit_arch=False)
    _load_initial_weights(object(), ma, explicit_arch=True)          # explicit architecture = declared random init
    _load_initial_weights(object(), ModelArgs(pretrained=False), explicit_arch=False)


def test_lr_horizon_follows_set_total_steps():
    """the schedule horizon must come from the dataset length, before the scheduler is built."""
    from contrastors_amd.trainers import TextTextTrainer

    class _T(TextTextTrainer):  # scheduler plumbing only: no device, no model
        def __init__(self, cfg):
            self.config, self.step, self.total_steps = cfg, 0, 10_000
            self.optimizer = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
            self.scheduler = self.get_scheduler(cfg, self.optimizer)

    cfg = Config(train_args=TrainArgs(warmup_pct=0.1, schedule_type="linear", learning_rate=1.0))
    t = _T(cfg)
    t.set_total_steps(100)
    lrs = []
    for _ in range(100):
        lrs.append(t.scheduler.get_last_lr()[0])
        t.optimizer.step()
        t.scheduler.step()
    assert lrs[0] == 0.0 and lrs[1] == pytest.approx(0.1) and abs(lrs[10] - 1.0) < 1e-6 and lrs[99] < 0.02


def test_selective_checkpointing_arena_and_config():
    """checkpoint_keep_layers: the arena of a checkpointed forward has 1 + k slots per intermediate buffer and L slots of the
    block-input tensor; the config accepts 'auto' / n and rejects the rest (train_args.checkpoint_keep_layers)."""
    import pytest as _pt
    from contrastors_amd.config import TrainArgs
    from contrastors_amd.nomic_bert import NomicBertConfig, _ChunkArena, parse_checkpoint_keep
    from contrastors_amd.vit import ViTConfig

    cfg = NomicBertConfig.nomic_bert_2048(vocab_size=512, n_layer=6, n_embd=128, n_head=2, n_inner=256)
    cpu = torch.device("cpu")
    full = _ChunkArena(cfg, 256, 6, True, 4, cpu)
    lit = _ChunkArena(cfg, 256, 6, True, 4, cpu, checkpoint=True)
    sel = _ChunkArena(cfg, 256, 6, True, 4, cpu, checkpoint=True, keep_layers=2)
    assert (full.desc.checkpoint, lit.desc.checkpoint, sel.desc.checkpoint) == (0, 1, 1)
    assert (full.desc.ckpt_keep, lit.desc.ckpt_keep, sel.desc.ckpt_keep) == (0, 0, 2)
    assert sel.tensors["qkv"].shape[0] == 3 and sel.tensors["yg"].shape[0] == 3 and sel.tensors["h1"].shape[0] == 3
    assert sel.tensors["h2"].shape[0] == 6 and lit.tensors["h2"].shape[0] == 6 and lit.tensors["qkv"].shape[0] == 1
    per_slot = _ChunkArena.slot_bytes_per_token(cfg) * 256
    assert sel.nbytes() - lit.nbytes() == 2 * per_slot
    assert lit.nbytes() < sel.nbytes() < full.nbytes()
    # a pre-norm (image) trunk keeps z1 per block
    vcfg = ViTConfig(n_embd=128, n_layer=4, n_head=2, n_inner=256, img_size=32, patch_size=16)
    v = _ChunkArena(vcfg, 128, 4, True, 4, cpu, checkpoint=True, keep_layers=9)   # clamped to n_layer
    assert v.desc.ckpt_keep == 4 and v.tensors["z1"].shape[0] == 4 and v.tensors["h2"].shape[0] == 5
    assert _ChunkArena(cfg, 256, 6, True, 4, cpu, checkpoint=False, keep_layers=3).desc.ckpt_keep == 0
    assert parse_checkpoint_keep(None, "x", default="auto") == "auto" and parse_checkpoint_keep("3", "x") == 3
    assert TrainArgs(checkpoint_keep_layers=0).checkpoint_keep_layers == 0 and TrainArgs().checkpoint_keep_layers == "auto"
    for bad in (-1, "some"):
        with _pt.raises(ValueError):
            TrainArgs(checkpoint_keep_layers=bad)


def test_selective_checkpointing_planner_budgets_the_measured_headroom_once(monkeypatch):
    """checkpoint_keep_layers = 'auto', host arithmetic only (torch.cuda.* mocked): after an arena's first, literal use the
    planner keeps as many blocks as 90 % of the device minus the step's measured peak pays for; a second arena of the same
    step sees what the first one was promised (per-device ledger), never more than the device has free right now; a
    suspended engine and an explicit keep count plan nothing; the ledger is settled when an arena dies."""
    import gc
    from types import SimpleNamespace

    from contrastors_amd import nomic_bert as nb

    GB = 1 << 30
    cfg = nb.NomicBertConfig.nomic_bert_2048(vocab_size=512, n_layer=12)
    per_tok = nb._ChunkArena.slot_bytes_per_token(cfg)
    assert per_tok == 2 * (3 * 768 + 768 + 3 * 768 + 3072 + 3072) + 4 * 16
    state = {"total": 288 * GB, "peak": 60 * GB, "free": 220 * GB, "reserved": 70 * GB, "allocated": 62 * GB}
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: SimpleNamespace(total_memory=state["total"]))
    monkeypatch.setattr(torch.cuda, "max_memory_allocated", lambda d=None: state["peak"])
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (state["free"], state["total"]))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda d=None: state["reserved"])
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda d=None: state["allocated"])
    dev = SimpleNamespace(index=7)   # a device index no other test touches
    eng = SimpleNamespace(config=cfg, device_=dev, CKPT_HBM_FRACTION=0.9, checkpoint_keep="auto", _keep_suspended=0,
                          _keep_plan={}, _keep_granted={}, _keep_granted_B={})
    eng._keep_mode = lambda: nb.NomicBertEngine._keep_mode(eng)
    plan = lambda arena: nb.NomicBertEngine._plan_keep(eng, arena)
    arena = lambda T, nbytes: SimpleNamespace(T_cap=T, probation=True, nbytes=lambda: nbytes)
    base = nb._hbm_grant(dev, 0)

    docs = arena(524288, 50 * GB)
    assert plan(docs) is True and docs.probation is False
    # 0.9 * 288 - 60 = 199.2 GB of headroom >= 12 blocks of 524288 tokens (12.0 GB each): everything is kept
    assert eng._keep_plan[524288] == 12 and eng._keep_granted[524288] == 12 * 524288 * per_tok
    queries = arena(65536, 6 * GB)
    assert plan(queries) is True and eng._keep_plan[65536] == 12          # 18 GB more still fit
    big = arena(1048576, 100 * GB)
    assert plan(big) is True
    granted = 12 * (524288 + 65536) * per_tok
    left = min(0.9 * 288 * GB - 60 * GB - granted,                        # what the ledger leaves of the measured headroom
               0.9 * (220 + 8 + 100) * GB - 100 * GB - granted)           # ... and of what the device has free right now
    assert eng._keep_plan[1048576] == int(left // (1048576 * per_tok)) == 1
    assert nb._hbm_grant(dev, 0) - base == sum(eng._keep_granted.values())
    # another process on the same GPU: the device's free memory, not this process's peak, is the binding bound
    state.update(free=20 * GB, reserved=62 * GB)
    eng._keep_plan.clear()
    tight = arena(262144, 25 * GB)
    assert plan(tight) is False and eng._keep_plan[262144] == 0           # 0.9 * (20 + 25) - 25 - grants < one block
    # suspended (resident GradCache) and explicit counts never plan
    eng._keep_suspended = 1
    assert plan(arena(131072, 10 * GB)) is False and 131072 not in eng._keep_plan
    eng._keep_suspended, eng.checkpoint_keep = 0, 3
    assert plan(arena(131072, 10 * GB)) is False and 131072 not in eng._keep_plan
    # an arena that dies returns its grant
    a = nb._ChunkArena(nb.NomicBertConfig.nomic_bert_2048(vocab_size=512, n_layer=2, n_embd=128, n_head=2, n_inner=256), 128, 2,
                       True, 2, torch.device("cpu"), checkpoint=True, keep_layers=1)
    a._device, a.granted = dev, 5 * GB
    before = nb._hbm_grant(dev, 5 * GB)
    del a
    gc.collect()
    assert nb._hbm_grant(dev, 0) == before - 5 * GB
    nb._HBM_GRANTED.pop(7, None)


def test_arena_pool_host_logic_on_cpu(monkeypatch):
    """NomicBertEngine._get_arena / release_arena without a GPU (arenas on the CPU device, planner mocked): size classes,
    best fit, a measured-and-dropped arena rebuilt at its own size, idle arenas given back, out-of-memory retry."""
    from types import SimpleNamespace

    from contrastors_amd import nomic_bert as nb

    E = nb.NomicBertEngine
    cfg = nb.NomicBertConfig.nomic_bert_2048(vocab_size=512, n_layer=2, n_embd=128, n_head=2, n_inner=256)
    eng = SimpleNamespace(config=cfg, device_=torch.device("cpu"), gradient_checkpointing=True, checkpoint_keep="auto",
                          _arena_free=[], _arena_nograd=None, _arena_tick=0, _keep_plan={}, _keep_granted={}, _keep_granted_B={}, _keep_suspended=0,
                          _keep_logged=set(), ARENA_IDLE_USES=4)
    for name in ("_keep_mode", "_checkpoint_keep_for", "_log_keep"):
        setattr(eng, name, (lambda n: lambda *a: getattr(E, n)(eng, *a))(name))

    def plan(arena):   # stands in for the measuring planner: every block fits
        arena.probation = False
        eng._keep_plan[arena.T_cap] = 2
        eng._keep_granted[arena.T_cap] = 1
        return True

    eng._plan_keep = plan
    get = lambda T: E._get_arena(eng, T, 4, True)
    rel = lambda a: E.release_arena(eng, a)

    a = get(3000)
    cap0 = a.T_cap
    assert cap0 == 3072 and a.probation and a.keep_layers == 0         # 128-token granules, then 64 size classes per octave
    rel(a)
    assert eng._arena_free == [] and eng._keep_plan == {cap0: 2}       # measured, dropped for its rebuild
    b = get(2500)
    assert b.T_cap == cap0 and b.keep_layers == 2 and not b.probation  # the smaller batch rebuilds the planned arena, at ITS size
    rel(b)
    c = get(5000)
    assert c.T_cap > cap0 and c.probation
    rel(c)
    d = get(4000)
    assert d.T_cap == c.T_cap and d.keep_layers == 2
    rel(d)
    assert sorted(x.T_cap for x in eng._arena_free) == [cap0, d.T_cap]
    assert get(2000) is b                                             # best fit, not first fit
    rel(b)
    eng._arena_tick += 10                                             # nobody took either arena for a while ...
    e = get(9000)                                                     # ... and a record batch comes: both are given back first
    assert e.probation and eng._arena_free == [] and cap0 not in eng._keep_plan and d.T_cap not in eng._keep_plan
    rel(e)
    # out of memory with an idle arena in the pool: the pool is emptied and the allocation tried again
    f = get(9000)
    rel(f)
    real, calls = nb._ChunkArena, []

    class Flaky(real):
        def __init__(self, *a, **k):
            calls.append(1)
            if len(calls) == 1:
                raise torch.OutOfMemoryError("simulated")
            super().__init__(*a, **k)

    monkeypatch.setattr(nb, "_ChunkArena", Flaky)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    eng.checkpoint_keep = 0
    g = get(20000)
    assert len(calls) == 2 and eng._arena_free == [] and g.T_cap >= 20000
    calls.clear()
    with pytest.raises(torch.OutOfMemoryError):
        get(40000)                                                    # nothing idle to give back: the caller's error


def test_compact_swiglu_backward_identity():
    """The algebra behind cx_gemm_bf16_swiglu_bwd_gate (the gated MLP keeps the gate alone): with act = y * silu(g),
    d gate = d * y * silu'(g) = d * act * (1 / g + 1 - sigmoid(g)).  Exact in fp64 against autograd of the reference's
    formula (flash_attn.ops.activations.swiglu: silu(x) * y, sc/layers/mlp.py:75); with act rounded to bf16 the recovered
    gradient carries that one rounding (2^-9 relative), as a saved bf16 y would."""
    g = torch.Generator().manual_seed(0)
    y = torch.randn(4096, 64, generator=g, dtype=torch.float64)
    gate = (torch.randn(4096, 64, generator=g, dtype=torch.float64) * 3).requires_grad_()
    d = torch.randn(4096, 64, generator=g, dtype=torch.float64)
    yv = y.clone().requires_grad_()
    act = torch.nn.functional.silu(gate) * yv
    act.backward(d)
    sg = torch.sigmoid(gate.detach())
    dgate = d * act.detach() * (1.0 / gate.detach() + 1.0 - sg)
    dy = d * gate.detach() * sg
    assert float((dgate - gate.grad).abs().max()) < 1e-11 and float((dy - yv.grad).abs().max()) < 1e-12
    act16 = act.detach().to(torch.bfloat16).double()
    dgate16 = d * act16 * (1.0 / gate.detach() + 1.0 - sg)
    rel = (dgate16 - gate.grad).abs() / gate.grad.abs().clamp_min(1e-300)
    assert float(rel.max()) <= 2.0 ** -8 and float(rel.mean()) < 2.0 ** -9


@pytest.mark.skipif(not REF_YAML.exists(), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("arch", ["nomic", "bert", "vit", "vit_clip"])
def test_decay_grouping_equals_the_reference_configure_optimizer(arch):
    """The engines lay their parameters out as [decay | no-decay] flat buffers (fused AdamW).  Which parameter
    goes where is the reference's rule (sc/optimizer.py:7-47: squeeze().ndim < 2, 'bias' in the name, logit_scale -> no
    decay): run the reference's OWN configure_optimizer on a module carrying the engine's parameter names and shapes and
    compare the two groups."""
    from types import SimpleNamespace

    from contrastors_amd import nomic_bert as nb
    from contrastors_amd.vit import ViTConfig, ViTEngine
    from oracle import ref_import

    ref_import.load()
    import importlib

    ref_opt = importlib.import_module("contrastors.optimizer")
    if arch in ("nomic", "bert"):
        cfg = (nb.NomicBertConfig.nomic_bert_2048(vocab_size=128, n_layer=2) if arch == "nomic"
               else nb.NomicBertConfig.bert_base_uncased(vocab_size=128, n_layer=2))
        cls = nb.NomicBertEngine
    else:
        cfg = ViTConfig(n_layer=2) if arch == "vit" else ViTConfig.clip_vit_base_patch16(n_layer=2)
        cls = ViTEngine
    fake = SimpleNamespace(config=cfg, _LAYER_PREFIX=cls._LAYER_PREFIX)
    fake._layer_specs = lambda l: cls._layer_specs(fake, l)
    decay, nodecay = cls._param_specs(fake)
    mod = torch.nn.Module()
    back = {}
    for name, shape in decay + nodecay:
        key = name.replace(".", "__")
        back[key] = name
        mod.register_parameter(key, torch.nn.Parameter(torch.zeros(*shape)))
    args = SimpleNamespace(weight_decay=0.01, learning_rate=1e-3, adam_beta1=0.9, adam_beta2=0.999, eps=1e-8)
    opt = ref_opt.configure_optimizer([mod], args)
    ids = {id(p): back[k] for k, p in mod.named_parameters()}
    ref_decay = {ids[id(p)] for p in opt.param_groups[0]["params"]}
    ref_nodecay = {ids[id(p)] for p in opt.param_groups[1]["params"]}
    assert opt.param_groups[0]["weight_decay"] == 0.01 and opt.param_groups[1]["weight_decay"] == 0.0
    assert ref_decay == {n for n, _ in decay}, ref_decay ^ {n for n, _ in decay}
    assert ref_nodecay == {n for n, _ in nodecay}, ref_nodecay ^ {n for n, _ in nodecay}


def test_contrastive_trainers_take_the_recipes_accumulation_steps():
    """gradient_accumulation_steps reaches the contrastive trainers' micro-step schedule (sc/trainers/base.py:366-393; the GPU
    test of the schedule itself is tests/test_trainer_gpu.py); nonsense values are refused."""
    from contrastors_amd.trainers import _accumulation_steps

    assert _accumulation_steps(TrainArgs()) == 1
    assert _accumulation_steps(TrainArgs(gradient_accumulation_steps=1)) == 1
    assert _accumulation_steps(TrainArgs(gradient_accumulation_steps=4)) == 4
    with pytest.raises(ValueError):
        _accumulation_steps(TrainArgs(gradient_accumulation_steps=0))


def test_reference_schema_keys_are_served_or_refused_never_dropped():
    """model_args.resid_pdrop reaches the text trunk's configuration (sc/config.py:187, modeling_biencoder.py:237); the keys of
    the reference schema this path does not serve are accepted at their inert defaults and refused otherwise."""
    from contrastors_amd.biencoder import BiEncoderConfig, trunk_config_with_overrides
    from contrastors_amd.config import ModelArgs
    from contrastors_amd.nomic_bert import NomicBertConfig
    from contrastors_amd.vit import ViTConfig

    ma = ModelArgs(resid_pdrop=0.1, ema=False, patch_dropout=0.0, num_experts=0)
    base = NomicBertConfig.nomic_bert_2048()
    got = trunk_config_with_overrides(BiEncoderConfig(resid_pdrop=ma.resid_pdrop), base)
    assert got.resid_pdrop == 0.1 and base.resid_pdrop == 0.0 and got.n_layer == base.n_layer
    assert trunk_config_with_overrides(BiEncoderConfig(), base) is base
    v = ViTConfig.vit_base_patch16_224()
    assert trunk_config_with_overrides(BiEncoderConfig(resid_pdrop=0.1), v) is v      # (image towers: not a text trunk)
    assert ModelArgs(ema=True).ema and ModelArgs(ema=True).ema_decay == 0.9999   # (round 4: served, sc/trainers/base.py:387-391)
    # (round 4: patch_dropout reaches the image trunk's configuration, modeling_biencoder.py:174,187; text trunks ignore it)
    assert trunk_config_with_overrides(BiEncoderConfig(patch_dropout=0.5), v).patch_dropout == 0.5 and v.patch_dropout == 0.0
    assert trunk_config_with_overrides(BiEncoderConfig(patch_dropout=0.5), base) is base
    for bad in (dict(ema_decay=1.5), dict(patch_dropout=1.0), dict(num_experts=8), dict(resid_pdrop=1.5)):
        with pytest.raises(ValueError):
            ModelArgs(**bad)


@pytest.mark.skipif(not REF_YAML.exists(), reason="reference tree only exists in the build container")
def test_every_reference_recipe_parses_or_is_refused_for_a_stated_reason():
    """All 17 recipes of sc/configs/train: the ones on the path parse unchanged; mixture-of-experts recipes are refused when the
    YAML is read (they used to parse with `num_experts` silently dropped, i.e. would have trained a dense model)."""
    import glob

    ok, refused = [], {}
    for f in sorted(glob.glob(str(REF_YAML.parent / "*.yaml"))):
        try:
            read_config(f)
            ok.append(Path(f).name)
        except ValueError as e:
            refused[Path(f).name] = str(e)
    assert set(refused) == {"contrastive_finetune_moe.yaml", "contrastive_pretrain_multilingual.yaml",
                            "contrastive_pretrain_multilingual_full.yaml", "contrastive_pretrain_tk2.yaml"}, refused
    assert all("num_experts" in v for v in refused.values())
    assert {"contrastive_pretrain.yaml", "contrastive_matryoshka.yaml", "mlm.yaml", "mmlm.yaml", "nomic_embed_vision_v1.5.yaml",
            "contrastive_finetune.yaml"} <= set(ok) and len(ok) == 13
    vis = read_config(str(REF_YAML.parent / "nomic_embed_vision_v1.5.yaml"))
    assert vis.vision_model_args.logit_scale == pytest.approx(1 / 0.07)   # `logit_scale: null` -> the default (sc/config.py:193-196)


@pytest.mark.parametrize("p", [0.15, 0.3])
def test_mask_tokens_is_the_collator_the_reference_uses(p):
    """sc/trainers/mlm.py:70,84 collates with transformers.DataCollatorForLanguageModeling: under the same torch RNG state
    mlm.mask_tokens must pick the same targets, the same [MASK] / random / kept split and the same random words -- bit for bit."""
    transformers = pytest.importorskip("transformers")
    from contrastors_amd.mlm import mask_tokens

    class Tok:   # the three things torch_mask_tokens asks its tokenizer
        mask_token, pad_token = "[MASK]", "[PAD]"

        def convert_tokens_to_ids(self, t):
            return 103

        def __len__(self):
            return 30522

    coll = transformers.DataCollatorForLanguageModeling(tokenizer=Tok(), mlm=True, mlm_probability=p)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1000, 30000, (32, 128), generator=g)
    special = torch.zeros_like(ids, dtype=torch.bool)
    special[:, 0] = True
    special[:, 100:] = True
    torch.manual_seed(5)
    want_ids, want_labels = coll.torch_mask_tokens(ids.clone(), special_tokens_mask=special.clone())
    torch.manual_seed(5)
    got_ids, got_labels = mask_tokens(ids, special, p, 103, 30522, None)
    assert torch.equal(got_labels, want_labels) and torch.equal(got_ids, want_ids)


def test_train_cli_overrides():
    """sc/train.py:50-94: `--key value` / `--key=value` / bare boolean flags land in every section that has the key, typed;
    unknown keys are an error (the reference's argparse rejects them; a typo must not train the YAML's value)."""
    from contrastors_amd.train import apply_overrides, parse_args, split_overrides

    args, ov = parse_args(["--config", "x.yaml", "--dtype", "bfloat16", "--local_rank", "3", "--learning_rate", "1e-3",
                           "--batch_size=64", "--gradient_checkpointing", "--seq_len", "512", "--weighted_sampling", "no",
                           "--resid_pdrop", "0.1", "--checkpoint-keep-layers", "auto", "--num_train_steps", "7"])
    assert args.config == "x.yaml" and args.dtype == "bfloat16" and args.local_rank == 3
    assert ov == {"learning_rate": "1e-3", "batch_size": "64", "gradient_checkpointing": True, "seq_len": "512",
                  "weighted_sampling": "no", "resid_pdrop": "0.1", "checkpoint_keep_layers": "auto", "num_train_steps": "7"}
    cfg = apply_overrides(Config(train_args=TrainArgs()), ov)
    assert cfg.train_args.learning_rate == 1e-3 and cfg.data_args.batch_size == 64 and cfg.model_args.seq_len == 512
    assert cfg.model_args.gradient_checkpointing is True and cfg.data_args.weighted_sampling is False
    assert cfg.model_args.resid_pdrop == 0.1 and cfg.train_args.checkpoint_keep_layers == "auto" and cfg.train_args.num_train_steps == 7
    with pytest.raises(SystemExit):
        apply_overrides(Config(train_args=TrainArgs()), {"learning_rte": "1"})
    with pytest.raises(ValueError):
        apply_overrides(Config(train_args=TrainArgs()), {"gradient_checkpointing": "maybe"})
    with pytest.raises(SystemExit):
        split_overrides(["stray"])


@pytest.mark.skipif(not REF_YAML.exists(), reason="reference tree only exists in the build container")
def test_oracle_pooling_head_equals_the_reference_modules():
    """oracle.encoder_ref.pool / the hamming LayerNorm / normalisation of `biencoder_embedding` against the reference's OWN
    modules (sc/models/biencoder/modeling_biencoder.py:44-49 ClsSelector, :79-90 MeanPooling incl. its mask-free ViT branch,
    :283 nn.LayerNorm(elementwise_affine=False), :317 F.normalize), imported by oracle/ref_import.load_biencoder()."""
    from oracle import encoder_ref, ref_import

    bi = ref_import.load_biencoder()
    g = torch.Generator().manual_seed(3)
    h = torch.randn(5, 9, 32, generator=g)
    lens = torch.tensor([9, 1, 4, 7, 2])
    mask = (torch.arange(9)[None] < lens[:, None]).long()
    ids = torch.zeros(5, 9, dtype=torch.long)
    assert torch.equal(encoder_ref.pool(h, mask, "cls"), bi.ClsSelector()(h, ids, mask))
    torch.testing.assert_close(encoder_ref.pool(h, mask, "mean"), bi.MeanPooling()(h, ids, mask), rtol=0, atol=0)
    torch.testing.assert_close(encoder_ref.pool(h, None, "mean"), bi.MeanPooling()(h, ids, None), rtol=0, atol=0)
    e = encoder_ref.pool(h, mask, "mean")
    ham = torch.nn.LayerNorm(32, elementwise_affine=False)
    torch.testing.assert_close(torch.nn.functional.layer_norm(e, (32,)), ham(e), rtol=0, atol=0)
    ls = bi.LogitScale(SimpleNamespace(logit_scale=20.0, trainable_logit_scale=True))
    assert float(ls(torch.ones(1)).detach()) == pytest.approx(20.0, rel=1e-6) and ls.logit_scale.requires_grad


@pytest.mark.skipif(not REF_YAML.exists(), reason="reference tree only exists in the build container")
def test_rand_context_behaves_like_the_reference_class_on_the_cpu_generator():
    """rand_state.RandContext against sc/rand_state.py:6-22 (the reference's own class, loaded from its file): numbers drawn
    inside `with ctx:` replay the stream that followed the snapshot, and the outer stream continues afterwards as if the
    block had not run; `needed=False` (dropout 0: the saving this port makes) leaves the generator alone entirely."""
    import importlib.util

    from contrastors_amd.rand_state import RandContext

    spec = importlib.util.spec_from_file_location("ref_rand_state", str(REF_YAML.parents[2] / "rand_state.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    def run(cls, **kw):
        torch.manual_seed(11)
        chunk = {"input_ids": torch.zeros(2, 3, dtype=torch.long)}
        torch.rand(3)                       # something before the snapshot
        ctx = cls(chunk, **kw)
        first = torch.rand(4)               # "pass 1" consumes numbers after the snapshot
        between = torch.rand(2)             # ... and the program goes on
        with ctx:
            replay = torch.rand(4)          # "pass 2" under the snapshot
        after = torch.rand(2)
        return first, between, replay, after

    a, b = run(ref.RandContext), run(RandContext)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(b[0], b[2])          # the replay IS pass 1's stream
    off = run(RandContext, needed=False)
    assert torch.equal(off[0], a[0]) and not torch.equal(off[2], off[0])   # no snapshot, no replay: the stream just continues


@pytest.mark.skipif(not REF_YAML.exists(), reason="reference tree only exists in the build container")
def test_saved_config_json_loads_with_the_reference_biencoder_config(tmp_path):
    """BiEncoder.save_pretrained's config.json through the reference's own BiEncoderConfig.from_pretrained
    (sc/models/biencoder/configuration_biencoder.py:4-31; what sc/trainers/text_text.py:148-159 does with `checkpoint:`)."""
    import importlib.util
    import json

    from contrastors_amd.biencoder import BiEncoderConfig, config_json
    from contrastors_amd.nomic_bert import NomicBertConfig

    spec = importlib.util.spec_from_file_location(
        "ref_bi_cfg", str(REF_YAML.parents[2] / "models" / "biencoder" / "configuration_biencoder.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    ours = BiEncoderConfig(model_name="nomic-ai/nomic-bert-2048", pooling="mean", logit_scale=50.0, hamming=True,
                           gradient_checkpointing=True, projection_dim=256, nomic_encoder=True, trainable_logit_scale=True)
    (tmp_path / "config.json").write_text(json.dumps(config_json(ours, NomicBertConfig.nomic_bert_2048())))
    got = ref.BiEncoderConfig.from_pretrained(str(tmp_path))
    for k in ("model_name", "pooling", "logit_scale", "hamming", "gradient_checkpointing", "projection_dim", "nomic_encoder",
              "trainable_logit_scale", "freeze", "pretrained", "use_fused_kernels"):
        assert getattr(got, k) == getattr(ours, k), k
    assert got.trunk_config["n_layer"] == 12 and got.trunk_type == "NomicBertConfig"


@pytest.mark.skipif(not REF_YAML.exists(), reason="reference tree only exists in the build container")
def test_bert_base_architecture_equals_the_reference_conversion_of_the_hf_config():
    """NomicBertConfig.bert_base_uncased(hf_dropout=True) against sc/models/encoder/bert.py:11-50 bert_config_to_nomic_config
    applied to transformers.BertConfig() (= bert-base-uncased's hub config): every field the engine reads, the dropout
    probabilities 0.1 the conversion carries over included; without the flag only those three differ."""
    import dataclasses
    import importlib.util
    import sys
    import types

    transformers = pytest.importorskip("transformers")
    from contrastors_amd.biencoder import _default_trunk_config
    from contrastors_amd.nomic_bert import NomicBertConfig

    enc = REF_YAML.parents[2] / "models" / "encoder"
    pkg = types.ModuleType("ref_enc_pkg")
    pkg.__path__ = [str(enc)]
    sys.modules["ref_enc_pkg"] = pkg
    mods = {}
    for n in ("configuration_nomic_bert", "bert"):
        spec = importlib.util.spec_from_file_location(f"ref_enc_pkg.{n}", str(enc / f"{n}.py"))
        mods[n] = importlib.util.module_from_spec(spec)
        sys.modules[f"ref_enc_pkg.{n}"] = mods[n]
        spec.loader.exec_module(mods[n])
    ref = mods["bert"].bert_config_to_nomic_config(transformers.BertConfig())
    ours = NomicBertConfig.bert_base_uncased(hf_dropout=True)
    skip = {"rotary_emb_base", "max_position_embeddings", "rotary_scaling_factor", "max_trained_positions"}  # (no rotary in BERT)
    diff = {f.name for f in dataclasses.fields(ours) if f.name not in skip and hasattr(ref, f.name)
            and getattr(ref, f.name) != getattr(ours, f.name)}
    assert diff == set(), diff
    assert (ours.resid_pdrop, ours.embd_pdrop, ours.attn_pdrop) == (0.1, 0.1, 0.1)
    plain = NomicBertConfig.bert_base_uncased()
    assert {f.name for f in dataclasses.fields(plain) if getattr(plain, f.name) != getattr(ours, f.name)} == {
        "resid_pdrop", "embd_pdrop", "attn_pdrop"}
    assert _default_trunk_config("bert-base-uncased") == ours


@pytest.mark.skipif(not REF_YAML.exists(), reason="reference tree only exists in the build container")
def test_vit_architectures_equal_the_reference_conversions_of_the_hf_configs():
    """ViTConfig.vit_base_patch16_224() against sc/models/vit/hf_vit.py:9-53 applied to transformers.ViTConfig()
    (google/vit-base-patch16-224) and ViTConfig.clip_vit_base_patch16() against sc/models/vit/clip.py:9-55 applied to the
    vision half of transformers.CLIPConfig with patch 16 (openai/clip-vit-base-patch16): every field of ours, equal."""
    import dataclasses
    import importlib.util

    transformers = pytest.importorskip("transformers")
    from contrastors_amd.vit import ViTConfig

    def load(name):
        spec = importlib.util.spec_from_file_location(f"ref_vit_{name}", str(REF_YAML.parents[2] / "models" / "vit" / f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    for ours, ref in ((ViTConfig.vit_base_patch16_224(), load("hf_vit").hf_vit_config_to_vit_config(transformers.ViTConfig())),
                      (ViTConfig.clip_vit_base_patch16(),
                       load("clip").clip_config_to_vit_config(transformers.CLIPConfig(vision_config={"patch_size": 16})))):
        diff = {f.name: (getattr(ref, f.name, "<absent>"), getattr(ours, f.name)) for f in dataclasses.fields(ours)
                if getattr(ref, f.name, "<absent>") != getattr(ours, f.name)}
        assert diff == {}, diff


def test_split_inputs_chunks_a_kept_tail_on_its_own():
    """loss._split_inputs(inputs, chunk, tail_seqs) (round 4, the partially resident GradCache schedule): the last tail_seqs
    sequences are chunked from the END -- whole chunks last, the shorter one in front of them -- and the region before them as
    usual; every sequence appears exactly once, in order, with its length."""
    import numpy as np
    import torch

    from contrastors_amd.loss import _split_inputs

    n, S = 1000, 8
    ids = torch.arange(n * S).view(n, S)
    lens = np.arange(n) % S + 1
    for chunk, tail in ((128, 0), (128, 300), (128, 1000), (128, 64), (300, 301), (64, 5000)):
        chunks = _split_inputs({"input_ids": ids, "seqlens": lens}, chunk, tail)
        sizes = [c["input_ids"].shape[0] for c in chunks]
        assert sum(sizes) == n and all(0 < z <= chunk for z in sizes)
        assert torch.equal(torch.cat([c["input_ids"] for c in chunks]), ids)
        assert np.array_equal(np.concatenate([c["seqlens"] for c in chunks]), lens)
        t = min(tail, n)
        k = (t + chunk - 1) // chunk                      # chunks of the kept region
        if k:
            assert sum(sizes[-k:]) == t
            assert all(z == chunk for z in sizes[len(sizes) - k + 1:]), (chunk, tail, sizes)   # whole chunks behind the short one
        head = sizes[: len(sizes) - k]
        assert all(z == chunk for z in head[:-1])          # the region before the tail: the usual chunking
