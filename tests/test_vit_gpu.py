"""ViT image tower (cx_vit_forward / cx_vit_backward: patchify, patch projection, cls + position embeddings, pre-norm
blocks, ln_f, pooling) vs (1) the golden fixture produced by the reference's own ViTModel python and (2) the fp32 oracle
at the ViT-B/16 architecture of BASELINE configs 4/5, judged with the reference's tolerance rule
err(new) <= 3 * err(bf16 eager) (tests/test_flash_vit.py:55-67)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from contrastors_amd.vit import ViTConfig, ViTEngine
from oracle import vit_ref
from tests.gpu_util import max_err, rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _oracle(sd, ns, pix, pooling, bf16):
    sdd = {k: v.detach().to(DEV).requires_grad_() for k, v in sd.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        emb = vit_ref.vit_embedding(sdd, ns, pix, pooling)
    return emb.float(), sdd


def _run(cfg, ns, sd, pix, pooling, probe):
    eng = ViTEngine(cfg, device=DEV, pooling=pooling)
    eng.load_reference_state_dict(sd)
    eng.train()
    emb, arena = eng.forward_chunk(pix, True)
    emb2, _ = eng.forward_chunk(pix, False)
    assert torch.equal(emb, emb2), "no-grad (single slot) forward must equal the saving forward"
    eng.zero_grad()
    eng.backward_chunk(pix, arena, probe)
    return emb, eng.reference_grad_dict()


@pytest.mark.parametrize("fixture", ["vit_tiny", "vit_clip_tiny"])
@pytest.mark.parametrize("pooling", ["cls", "mean"])
def test_vit_matches_reference_golden(gold, pooling, fixture):
    """vit_clip_tiny: the OpenAI-CLIP flavour of the tower (quick_gelu in the fused fc1 epilogue and its backward, the
    pre-LayerNorm ahead of the first block, no patch-embedding bias; sc/models/vit/clip.py:14-58)."""
    g = gold(fixture)
    d = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg/")}
    cfg, ns = ViTConfig(**d), SimpleNamespace(**d)
    sd = vit_ref.random_state_dict(ns, int(g["seed"]))
    pix = torch.from_numpy(g["pixels"]).to(DEV)
    probe = torch.from_numpy(g[f"{pooling}/probe"]).to(DEV)
    emb, grads = _run(cfg, ns, sd, pix, pooling, probe)
    gold_emb = torch.from_numpy(g[f"{pooling}/embedding"]).to(DEV)
    emb16, sd16 = _oracle(sd, ns, pix, pooling, True)
    (emb16 * probe).sum().backward()
    e_hip, e_b = max_err(emb, gold_emb), max_err(emb16, gold_emb)
    worst = 0.0
    for k in g.files:
        if not k.startswith(f"{pooling}/gnorm/"):
            continue
        n = k[len(pooling) + 7:]
        want, got, bf = float(g[k]), float(grads[n].norm()), float(sd16[n].grad.norm())
        worst = max(worst, abs(got - want) / max(want, 1e-6))
        assert abs(got - want) <= 3 * abs(bf - want) + 2e-2 * want + 1e-5, f"{n}: |grad| {got} vs reference {want} (bf16 {bf})"
    errs = {}
    for n, sl in (("embeddings.cls_token", None), ("embeddings.pos_embed", None), ("ln_f.weight", None),
                  ("embeddings.proj.weight", 16), ("layers.0.attn.Wqkv.weight", 16), ("layers.1.mlp.fc2.weight", 16)):
        got = grads[n] if sl is None else grads[n][:sl, :sl]
        key = f"{pooling}/g/{n}" + ("" if sl is None else "[:16,:16]")
        errs[n] = rel_err(got.reshape(-1), torch.from_numpy(g[key]).to(DEV).reshape(-1))
    report("vit_golden", pooling=pooling, e_emb_hip=e_hip, e_emb_bf16=e_b, worst_gnorm_rel=worst, **errs)
    assert e_hip <= 5e-3 and e_hip <= 3 * e_b + 1e-4
    assert all(v < 5e-2 for v in errs.values()), errs


def test_vit_b16_vs_oracle():
    """ViT-B/16 at 224x224 (197 tokens per image, T not a multiple of 64), B = 3, fp32 oracle as judge."""
    cfg = ViTConfig.vit_base_patch16_224(n_layer=4)  # 4 of the 12 identical blocks: oracle runtime
    ns = SimpleNamespace(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    sd = vit_ref.random_state_dict(ns, 9)
    g = torch.Generator().manual_seed(10)
    pix = torch.randn(3, 3, 224, 224, generator=g).to(DEV)
    probe = torch.randn(3, cfg.n_embd, generator=g).to(DEV)
    emb, grads = _run(cfg, ns, sd, pix, "cls", probe)
    ref, sd32 = _oracle(sd, ns, pix, "cls", False)
    ref16, sd16 = _oracle(sd, ns, pix, "cls", True)
    (ref * probe).sum().backward()
    (ref16 * probe).sum().backward()
    e_hip, e_b = max_err(emb, ref), max_err(ref16, ref)
    worst_ratio, worst_name = 0.0, ""
    table = []
    for n, gh in grads.items():
        eh, eb = rel_err(gh.reshape(-1), sd32[n].grad.reshape(-1)), rel_err(sd16[n].grad.float().reshape(-1), sd32[n].grad.reshape(-1))
        table.append((eh / (eb + 1e-4), n, eh, eb))
        if eh / (eb + 1e-4) > worst_ratio:
            worst_ratio, worst_name = eh / (eb + 1e-4), n
        # ViT-B/16 blocks: the 3 x bf16-eager rule with no additive floor (1e-4 only guards eb = 0)
        assert eh <= 3 * (eb + 1e-4), f"{n}: rel grad err {eh:.4f} vs bf16 eager {eb:.4f}"
    table.sort(reverse=True)
    report("vit_b16", e_emb_hip=e_hip, e_emb_bf16=e_b, worst_grad_ratio=worst_ratio, worst_grad_name=worst_name,
           worst_ratios="; ".join(f"{n} {eh:.4f}/{eb:.4f}" for _, n, eh, eb in table[:6]))
    assert e_hip <= 3 * e_b + 1e-4


def test_vit_bf16_pixels_and_errors():
    cfg = ViTConfig(n_embd=256, n_layer=1, n_head=4, n_inner=512, img_size=32, patch_size=8)
    eng = ViTEngine(cfg, device=DEV, seed=1).eval()
    pix = torch.randn(4, 3, 32, 32, device=DEV)
    a = eng(pix)
    b = eng(pix.to(torch.bfloat16))
    assert max_err(a, b) < 2e-2
    with pytest.raises(ValueError):
        eng(torch.randn(4, 3, 64, 64, device=DEV))
    with pytest.raises(NotImplementedError):
        ViTConfig(resid_pdrop=0.1)


def test_dual_encoder_native_towers_clip_step():
    """BASELINE config 5 in miniature: CLIP-style DualEncoder with a native ViT image tower and a native BERT-family
    text tower, both trained (sc/models/dual_encoder/modeling_dual_encoder.py:36-68), vs the fp32 oracle towers."""
    from contrastors_amd.biencoder import BiEncoder, BiEncoderConfig, DualEncoder, LogitScale
    from contrastors_amd.nomic_bert import NomicBertConfig
    from oracle import encoder_ref

    vcfg = ViTConfig(n_embd=256, n_layer=2, n_head=4, n_inner=512, img_size=32, patch_size=8, layer_norm_epsilon=1e-6)
    tcfg = NomicBertConfig.bert_base_uncased(vocab_size=512, n_embd=256, n_layer=2, n_head=4, n_inner=512,
                                             max_position_embeddings=64)
    vns = SimpleNamespace(**{k: getattr(vcfg, k) for k in vcfg.__dataclass_fields__})
    tns = SimpleNamespace(**{k: getattr(tcfg, k) for k in tcfg.__dataclass_fields__})
    vsd, tsd = vit_ref.random_state_dict(vns, 21), encoder_ref.random_state_dict(tns, 22)
    vision = BiEncoder(BiEncoderConfig(pooling="cls", trunk_config=vcfg), device=DEV)
    text = BiEncoder(BiEncoderConfig(pooling="mean", trunk_config=tcfg), device=DEV)
    vision.trunk.load_reference_state_dict(vsd)
    text.trunk.load_reference_state_dict(tsd)
    de = DualEncoder(text, vision, LogitScale(SimpleNamespace(logit_scale=20.0, trainable_logit_scale=False)).to(DEV)).train()
    g = torch.Generator().manual_seed(23)
    n = 8
    pix = torch.randn(n, 3, 32, 32, generator=g).to(DEV)
    ids = torch.randint(3, 512, (n, 16), generator=g).to(DEV)
    mask = torch.ones(n, 16, dtype=torch.long, device=DEV)
    vision.trunk.zero_grad()
    text.trunk.zero_grad()
    out = de({"input_ids": ids, "attention_mask": mask}, {"input_ids": pix})
    out["loss"].backward()
    # oracle: same towers in fp32, symmetric InfoNCE
    vs = {k: v.to(DEV).requires_grad_() for k, v in vsd.items()}
    ts = {k: v.to(DEV).requires_grad_() for k, v in tsd.items()}
    ve = vit_ref.vit_embedding(vs, vns, pix, "cls")
    te = encoder_ref.biencoder_embedding(ts, tns, ids, mask)
    labels = torch.arange(n, device=DEV)
    ref = 0.5 * (torch.nn.functional.cross_entropy(20.0 * ve @ te.T, labels)
                 + torch.nn.functional.cross_entropy(20.0 * te @ ve.T, labels))
    ref.backward()
    e_loss = abs(float(out["loss"].detach()) - float(ref.detach()))
    gv, gt = vision.trunk.reference_grad_dict(), text.trunk.reference_grad_dict()
    e_v = rel_err(gv["embeddings.proj.weight"], vs["embeddings.proj.weight"].grad)
    e_t = rel_err(gt["encoder.layers.0.attn.Wqkv.weight"], ts["encoder.layers.0.attn.Wqkv.weight"].grad)
    report("dual_encoder_native", loss=float(out["loss"]), ref=float(ref), e_loss=e_loss, e_vproj=e_v, e_tqkv=e_t)
    assert e_loss < 2e-2 and e_v < 5e-2 and e_t < 5e-2


def test_dual_encoder_precomputed_text_lit():
    """LiT with offline text embeddings (modeling_dual_encoder.py:13-18,37-41): the frozen text tower is never run, the
    batch's `text_embs` are used; same loss / image-tower gradient as running the frozen tower on the same texts."""
    from contrastors_amd.biencoder import BiEncoder, BiEncoderConfig, DualEncoder, LogitScale
    from contrastors_amd.nomic_bert import NomicBertConfig

    vcfg = ViTConfig(n_embd=256, n_layer=2, n_head=4, n_inner=512, img_size=32, patch_size=8, layer_norm_epsilon=1e-6)
    tcfg = NomicBertConfig.bert_base_uncased(vocab_size=512, n_embd=256, n_layer=2, n_head=4, n_inner=512,
                                             max_position_embeddings=64)
    vision = BiEncoder(BiEncoderConfig(pooling="cls", trunk_config=vcfg), device=DEV, seed=1).train()
    text = BiEncoder(BiEncoderConfig(pooling="mean", trunk_config=tcfg, freeze=True), device=DEV, seed=2)
    scale = LogitScale(SimpleNamespace(logit_scale=20.0, trainable_logit_scale=False)).to(DEV)
    with pytest.raises(AssertionError):
        DualEncoder(BiEncoder(BiEncoderConfig(pooling="mean", trunk_config=tcfg), device=DEV), vision, scale,
                    precomputed_text=True)
    g = torch.Generator().manual_seed(5)
    n = 8
    pix = torch.randn(n, 3, 32, 32, generator=g).to(DEV)
    ids = torch.randint(3, 512, (n, 16), generator=g).to(DEV)
    mask = torch.ones(n, 16, dtype=torch.long, device=DEV)
    live = DualEncoder(text, vision, scale).train()
    vision.trunk.zero_grad()
    a = live({"input_ids": ids, "attention_mask": mask}, {"input_ids": pix})
    a["loss"].backward()
    g_live = vision.trunk.flat_grad.clone()
    with torch.no_grad():
        embs = live.encode_text({"input_ids": ids, "attention_mask": mask}, normalize=False)
    pre = DualEncoder(text, vision, scale, precomputed_text=True).train()
    with pytest.raises(AssertionError):
        pre({"input_ids": ids}, {"input_ids": pix})
    vision.trunk.zero_grad()
    b = pre({"text_embs": embs.cpu()}, {"input_ids": pix})
    b["loss"].backward()
    assert abs(float(a["loss"].detach()) - float(b["loss"].detach())) < 1e-5
    assert float((vision.trunk.flat_grad - g_live).norm()) <= 1e-4 * float(g_live.norm())
    assert float(text.trunk.flat_grad.abs().max()) == 0.0
    assert live.encode_image(pix).shape == (n, 256)



def test_map_pooling_head_on_the_vit_tower_matches_the_oracle(gold):
    """`pooling: map` (configs/train/nomic_embed_vision_v1.5.yaml:69): BiEncoder(ViT) + the attention
    pooling head.  (1) The head alone on the reference-generated golden's hidden states vs the reference's own output
    (oracle/map_pool_ref.py is pinned to the same file on CPU); (2) the whole tower -- native ViT hidden states, head on the
    HIP GEMM / K3 attention kernels, gradient back through cx_vit_backward_hidden -- vs the fp32 oracle composition, with
    the 3 x bf16-eager rule."""
    from contrastors_amd.biencoder import BiEncoder, BiEncoderConfig
    from contrastors_amd.map_pooling import MultiHeadAttentionPooling
    from oracle import map_pool_ref

    g = gold("map_pool_tiny")
    d, inner, H, eps = int(g["d"]), int(g["inner"]), int(g["n_head"]), float(g["eps"])
    sd_head = map_pool_ref.random_state_dict(d, inner, int(g["seed"]))
    hc = SimpleNamespace(n_embd=d, n_head=H, n_inner=inner, layer_norm_epsilon=eps, activation_function="gelu",
                         qkv_proj_bias=True, mlp_fc1_bias=True, mlp_fc2_bias=True, use_rms_norm=False)
    head = MultiHeadAttentionPooling(hc, device=DEV)
    head.load_state_dict(sd_head)
    hid = torch.from_numpy(g["hidden"]).to(DEV)
    h16 = hid.to(torch.bfloat16).requires_grad_()
    out = head(h16, None, None)
    want = torch.from_numpy(g["out"]).to(DEV)
    assert max_err(out, want) < 3e-2 and rel_err(out, want) < 8e-3, "bf16 hidden states and GEMMs vs the fp32 reference"
    (out * torch.from_numpy(g["probe"]).to(DEV)).sum().backward()
    assert rel_err(h16.grad.float(), torch.from_numpy(g["g/hidden"]).to(DEV)) < 2e-2
    for k in g.files:
        if k.startswith("gnorm/"):
            n = k[6:]
            got = dict(head.named_parameters())[n].grad
            assert abs(float(got.norm()) - float(g[k])) <= 3e-2 * float(g[k]) + 1e-6, n

    # (2) the whole tower
    vg = gold("vit_tiny")
    dd = {k[4:]: vg[k].item() for k in vg.files if k.startswith("cfg/")}
    cfg, ns = ViTConfig(**dd), SimpleNamespace(**dd)
    assert cfg.n_embd == d and cfg.n_head == H
    sd = vit_ref.random_state_dict(ns, int(vg["seed"]))
    tower = BiEncoder(BiEncoderConfig(model_name="vit", pooling="map", trunk_config=cfg), device=DEV, seed=4).train()
    tower.trunk.load_reference_state_dict(sd)
    tower.selector.load_state_dict(sd_head)
    pix = torch.from_numpy(vg["pixels"]).to(DEV)
    probe = torch.randn(pix.shape[0], d, generator=torch.Generator().manual_seed(3)).to(DEV)
    tower.trunk.zero_grad()
    emb = tower(input_ids=pix)["embedding"]
    (emb * probe).sum().backward()
    assert tower.trunk._outstanding == 0

    def oracle(bf16):
        sdd = {k: v.detach().to(DEV).requires_grad_() for k, v in sd.items()}
        shd = {k: v.detach().to(DEV).requires_grad_() for k, v in sd_head.items()}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            hidden = vit_ref.vit_hidden(sdd, ns, pix)
            e = torch.nn.functional.normalize(map_pool_ref.map_pool(shd, hidden.float() if not bf16 else hidden, H, eps).float(), dim=-1)
        (e.float() * probe).sum().backward()
        return e.float(), sdd, shd

    ref, sd32, sh32 = oracle(False)
    ref16, sd16, sh16 = oracle(True)
    e_hip, e_b = max_err(emb, ref), max_err(ref16, ref)
    assert e_hip <= 3 * e_b + 1e-4 and e_hip < 5e-3, (e_hip, e_b)
    grads = tower.trunk.reference_grad_dict()
    worst = 0.0
    for n, gh in grads.items():
        eh, eb = rel_err(gh.reshape(-1), sd32[n].grad.reshape(-1)), rel_err(sd16[n].grad.float().reshape(-1), sd32[n].grad.reshape(-1))
        worst = max(worst, eh / (eb + 1e-4))
        assert eh <= 3 * eb + 1e-2, f"{n}: rel grad err {eh:.4f} vs bf16 eager {eb:.4f}"
    for n, p_ in tower.selector.named_parameters():
        eh, eb = rel_err(p_.grad.reshape(-1), sh32[n].grad.reshape(-1)), rel_err(sh16[n].grad.float().reshape(-1), sh32[n].grad.reshape(-1))
        worst = max(worst, eh / (eb + 1e-4))
        assert eh <= 3 * eb + 1e-2, f"selector.{n}: rel grad err {eh:.4f} vs bf16 eager {eb:.4f}"
    report("vit_map_pooling", e_emb_hip=e_hip, e_emb_bf16=e_b, worst_grad_ratio=worst)
    # text towers and decoder-only poolings say so instead of silently doing something else
    from contrastors_amd.nomic_bert import NomicBertConfig

    with pytest.raises(NotImplementedError):
        BiEncoder(BiEncoderConfig(model_name="t", pooling="map", trunk_config=NomicBertConfig.nomic_bert_2048(n_layer=1)), device=DEV)
    with pytest.raises(NotImplementedError):
        BiEncoder(BiEncoderConfig(model_name="t", pooling="last", trunk_config=cfg), device=DEV)


@pytest.mark.parametrize("pooling", ["cls", "mean"])
def test_vit_patch_dropout_matches_reference_golden(gold, pooling):
    """PatchDropout (sc/layers/embedding.py:415-418, 519-557; round 4): in training every image keeps [cls] + the top-k of a CPU
    standard-normal draw over its patches.  The engine draws from the same generator in the same way, gathers ONLY the kept
    patches (patchify -> projection -> blocks on K + 1 tokens) and scatters the position gradients back to the original
    positions.  Judged against vit_patchdrop_tiny.npz = the reference's own ViTModel in training mode with patch_dropout 0.5."""
    g = gold("vit_patchdrop_tiny")
    d = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg/")}
    cfg, ns = ViTConfig(**d, patch_dropout=0.5), SimpleNamespace(**d)
    sd = vit_ref.random_state_dict(ns, int(g["seed"]))
    pix = torch.from_numpy(g["pixels"]).to(DEV)
    probe = torch.from_numpy(g[f"{pooling}/probe"]).to(DEV)
    keep = torch.from_numpy(g["keep"])
    eng = ViTEngine(cfg, device=DEV, pooling=pooling)
    eng.load_reference_state_dict(sd)
    eng.train()
    torch.manual_seed(int(g["rng_seed"]))
    emb, arena = eng.forward_chunk(pix, True)
    assert torch.equal(arena.patch_subset[0].cpu().long(), keep), "the engine's draw must select the reference's patches"
    eng.zero_grad()
    eng.backward_chunk(pix, arena, probe)
    grads = eng.reference_grad_dict()
    gold_emb = torch.from_numpy(g[f"{pooling}/embedding"]).to(DEV)
    sdd = {k: v.detach().to(DEV).requires_grad_() for k, v in sd.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        emb16 = vit_ref.vit_embedding(sdd, ns, pix, pooling, keep=keep).float()
    (emb16 * probe).sum().backward()
    e_hip, e_b = max_err(emb, gold_emb), max_err(emb16, gold_emb)
    for k in g.files:
        if not k.startswith(f"{pooling}/gnorm/"):
            continue
        n = k[len(pooling) + 7:]
        want, got, bf = float(g[k]), float(grads[n].norm()), float(sdd[n].grad.norm())
        assert abs(got - want) <= 3 * abs(bf - want) + 2e-2 * want + 1e-5, f"{n}: |grad| {got} vs reference {want} (bf16 {bf})"
    e_pos = rel_err(grads["embeddings.pos_embed"].reshape(-1), torch.from_numpy(g[f"{pooling}/g/embeddings.pos_embed"]).to(DEV).reshape(-1))
    e_cls = rel_err(grads["embeddings.cls_token"].reshape(-1), torch.from_numpy(g[f"{pooling}/g/embeddings.cls_token"]).to(DEV).reshape(-1))
    report("vit_patch_dropout", pooling=pooling, e_emb_hip=e_hip, e_emb_bf16=e_b, e_pos=e_pos, e_cls=e_cls)
    assert e_hip <= 5e-3 and e_hip <= 3 * e_b + 1e-4
    assert e_pos < 5e-2 and e_cls < 5e-2
    # a dropped patch's position receives no gradient from the images that dropped it: positions nobody kept are exactly zero
    gp = grads["embeddings.pos_embed"].reshape(-1, cfg.n_embd)
    kept_any = torch.zeros(cfg.n_patch, dtype=torch.bool)
    kept_any[keep.reshape(-1)] = True
    for pi in range(cfg.n_patch):
        if not kept_any[pi]:
            assert float(gp[1 + pi].abs().max()) == 0.0
    # eval mode: no dropout, the full sequence
    eng.eval()
    full, _ = eng.forward_chunk(pix, False)
    ref_full = vit_ref.vit_embedding({k: v.to(DEV) for k, v in sd.items()}, ns, pix, pooling)
    assert max_err(full, ref_full) <= 5e-3


def test_gradcache_replays_the_patch_dropout_draw():
    """PatchDropout draws from torch's CPU generator.  GradCache's pass 2 re-forwards every chunk, and the cached
    embedding gradients belong to the patch subsets pass 1 drew -- the per-chunk RandContext has to replay the draw although
    every dropout PROBABILITY of the image tower is 0 (ViTEngine.uses_rng).  Two-pass grad_cache_loss (resident=False) on an
    image tower (queries) and a text tower (documents) with patch_dropout = 0.5 against the direct loss on the same draws."""
    from contrastors_amd.biencoder import BiEncoder, BiEncoderConfig, LogitScale
    from contrastors_amd.loss import clip_loss, grad_cache_loss
    from contrastors_amd.nomic_bert import NomicBertConfig
    from contrastors_amd.policy import GradCachePolicy
    from oracle.make_golden import TINY_NOMIC

    vc = ViTConfig(n_embd=256, n_layer=2, n_head=4, n_inner=512, img_size=32, patch_size=8, patch_dropout=0.5)
    tc = NomicBertConfig(**{k: v for k, v in TINY_NOMIC.items() if k in NomicBertConfig.__dataclass_fields__})
    vis = BiEncoder(BiEncoderConfig(model_name="v", pooling="cls", trunk_config=vc), device=DEV, seed=1).train()
    txt = BiEncoder(BiEncoderConfig(model_name="t", pooling="mean", trunk_config=tc), device=DEV, seed=2).train()
    assert vis.trunk.uses_rng and not txt.trunk.uses_rng
    scale = LogitScale(SimpleNamespace(logit_scale=20.0, trainable_logit_scale=False)).to(DEV)
    g = torch.Generator().manual_seed(5)
    pix = {"input_ids": torch.randn(16, 3, 32, 32, generator=g).to(DEV)}
    doc = {"input_ids": torch.randint(3, 512, (16, 24), generator=g).to(DEV), "attention_mask": torch.ones(16, 24, dtype=torch.long, device=DEV)}

    def direct(chunk):
        # the direct loss with the SAME draws GradCache's pass 1 makes: one CPU randn per chunk of `chunk` images, in chunk order
        torch.manual_seed(77)
        vis.trunk.zero_grad(); txt.trunk.zero_grad()
        qs = [vis(input_ids=pix["input_ids"][a:a + chunk])["embedding"] for a in range(0, 16, chunk)]
        loss = clip_loss(torch.cat(qs), txt(**doc)["embedding"], scale)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), vis.trunk.flat_grad.clone(), txt.trunk.flat_grad.clone()

    l_ref, gv_ref, gt_ref = direct(4)
    torch.manual_seed(77)
    vis.trunk.zero_grad(); txt.trunk.zero_grad()
    loss = grad_cache_loss(vis, pix, txt, doc, chunk_size=4, logit_scale=scale, policy=GradCachePolicy(chunk="exact", resident=False))
    torch.cuda.synchronize()
    assert abs(float(loss) - l_ref) <= 1e-5 * abs(l_ref), (float(loss), l_ref)
    ev = float((vis.trunk.flat_grad - gv_ref).norm() / gv_ref.norm())
    et = float((txt.trunk.flat_grad - gt_ref).norm() / gt_ref.norm())
    report("gradcache_patch_dropout", e_vision=ev, e_text=et)
    # same kernels on the same chunks: only the accumulation order of the chunks' gradients differs.  A re-forward on ANOTHER
    # patch subset (the bug) gives gradients of a different function: e_vision ~ 1
    assert ev < 2e-3 and et < 2e-3, (ev, et)
