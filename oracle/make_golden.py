"""Generate tests/golden/*.npz by running the reference's own code on CPU (fp32).  TEST INFRASTRUCTURE.

Run in the build container:  python -m oracle.make_golden
Fixtures are small (inputs + outputs only); model weights are re-created from a seed by
oracle.encoder_ref.random_state_dict and pinned by a checksum stored in the fixture.
"""
from __future__ import annotations

import json
import os
import sys
import warnings
from pathlib import Path
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import encoder_ref, ref_import, vit_ref  # noqa: E402

GOLD = ROOT / "tests" / "golden"

TINY_NOMIC = dict(vocab_size=512, n_positions=64, n_embd=256, n_layer=2, n_head=4, n_inner=512,
                  activation_function="swiglu", rotary_emb_fraction=1.0, rotary_emb_base=1000.0,
                  qkv_proj_bias=False, mlp_fc1_bias=False, mlp_fc2_bias=False, layer_norm_epsilon=1e-12,
                  type_vocab_size=2, pad_token_id=0, max_position_embeddings=64)
TINY_BERT = dict(TINY_NOMIC, activation_function="gelu", rotary_emb_fraction=0.0, qkv_proj_bias=True,
                 mlp_fc1_bias=True, mlp_fc2_bias=True)


def cfg_ns(d):
    return SimpleNamespace(**d)


def ref_model(cfgd, sd):
    _, rcfg, rmod = ref_import.load()
    c = rcfg.NomicBertConfig(
        vocab_size=cfgd["vocab_size"], n_positions=cfgd["n_positions"], n_embd=cfgd["n_embd"],
        n_layer=cfgd["n_layer"], n_head=cfgd["n_head"], n_inner=cfgd["n_inner"],
        activation_function=cfgd["activation_function"], rotary_emb_fraction=cfgd["rotary_emb_fraction"],
        rotary_emb_base=cfgd["rotary_emb_base"], qkv_proj_bias=cfgd["qkv_proj_bias"],
        mlp_fc1_bias=cfgd["mlp_fc1_bias"], mlp_fc2_bias=cfgd["mlp_fc2_bias"],
        layer_norm_epsilon=cfgd["layer_norm_epsilon"], type_vocab_size=cfgd["type_vocab_size"],
        pad_token_id=cfgd["pad_token_id"], resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
        max_position_embeddings=cfgd["max_position_embeddings"], prenorm=False, use_flash_attn=False,
        rotary_scaling_factor=cfgd.get("rotary_scaling_factor"), max_trained_positions=cfgd.get("max_trained_positions", 2048),
    )
    m = rmod.NomicBertModel(c, add_pooling_layer=False)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("inv_freq" in k or "norm_factor" in k for k in missing), missing
    return m.eval()


def checksum(sd):
    return np.array([float(sum(v.double().sum() for v in sd.values())),
                     float(sum((v.double() ** 2).sum() for v in sd.values()))])


def make_inputs(cfgd, B, S, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, cfgd["vocab_size"], (B, S), generator=g)
    lens = torch.randint(S // 2, S + 1, (B,), generator=g)
    lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).long()
    ids = ids * mask  # pad id 0
    return ids, mask, lens


def gen_encoder(name, cfgd, seed):
    cfg = cfg_ns(cfgd)
    sd = encoder_ref.random_state_dict(cfg, seed)
    m = ref_model(cfgd, sd)
    ids, mask, lens = make_inputs(cfgd, 6, 32, seed + 1)
    for p in m.parameters():
        p.requires_grad_(True)
    hid = m(ids, attention_mask=mask).last_hidden_state
    hid0 = hid * mask.unsqueeze(-1)  # reference tests zero the padded rows (tests/test_flash_bert.py:65,71)
    # BiEncoder pooling restated (modeling_biencoder.py:79-90, :317) on top of the REFERENCE hidden states
    s = (hid * mask.unsqueeze(-1).float()).sum(1) / mask.sum(1, keepdim=True).float()
    emb = torch.nn.functional.normalize(s, dim=-1)
    g = torch.Generator().manual_seed(seed + 2)
    probe = torch.randn(emb.shape, generator=g)
    (emb * probe).sum().backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    keep = {}
    for k, gten in grads.items():
        keep["gnorm/" + k] = np.array(float(gten.norm()))
    keep["g/emb_ln.weight"] = grads["emb_ln.weight"].numpy()
    keep["g/encoder.layers.0.attn.Wqkv.weight[:16,:16]"] = grads["encoder.layers.0.attn.Wqkv.weight"][:16, :16].numpy()
    keep["g/encoder.layers.1.mlp.fc2.weight[:16,:16]"] = grads["encoder.layers.1.mlp.fc2.weight"][:16, :16].numpy()
    keep["g/embeddings.word_embeddings.weight[rows]"] = grads["embeddings.word_embeddings.weight"][ids[0, :8]].numpy()
    np.savez_compressed(GOLD / f"{name}.npz", seed=seed, input_ids=ids.numpy(), attention_mask=mask.numpy(),
                        lens=lens.numpy(), hidden=hid0.detach().numpy(), embedding=emb.detach().numpy(),
                        probe=probe.numpy(), weight_checksum=checksum(sd), **keep,
                        **{"cfg/" + k: np.array(v) for k, v in cfgd.items()})
    print(name, "hidden", tuple(hid.shape), "emb norm", float(emb.norm()))


TINY_VIT = dict(n_embd=256, n_layer=2, n_head=4, n_inner=512, img_size=32, patch_size=8, num_channels=3,
                layer_norm_epsilon=1e-6)
TINY_VIT_CLIP = dict(TINY_VIT, layer_norm_epsilon=1e-5, activation_function="quick_gelu", prepre_layernom=True,
                     patch_embed_bias=False)


def gen_vit(name, cfgd, seed, patch_dropout=0.0):
    """tests/golden/vit_tiny.npz: the reference ViTModel (sc/models/vit/vit.py) run on CPU in fp32.
    patch_dropout > 0 (vit_patchdrop_tiny.npz): the model in TRAINING mode with the reference's PatchDropout
    (sc/layers/embedding.py:415-418, 519-557) drawing from torch's CPU generator seeded with `rng_seed` right before the forward."""
    from transformers import GPT2Config

    vit = ref_import.load_vit()
    cfg = cfg_ns(cfgd)
    clip = bool(getattr(cfg, "prepre_layernom", False))   # the OpenAI-CLIP flavour (sc/models/vit/clip.py:14-58)
    c = GPT2Config(
        n_embd=cfg.n_embd, n_layer=cfg.n_layer, n_head=cfg.n_head, n_inner=cfg.n_inner,
        activation_function=getattr(cfg, "activation_function", "gelu"),
        vocab_size=0, n_positions=0, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
        layer_norm_epsilon=cfg.layer_norm_epsilon, initializer_range=0.02, bos_token_id=None, eos_token_id=None,
        drop_path_rate=0.0, prepre_layernom=clip, layer_scale=False, layer_scale_init=None, img_size=cfg.img_size,
        patch_size=cfg.patch_size, num_channels=cfg.num_channels, prenorm=True, parallel_block=False,
        parallel_block_tied_norm=False, rotary_emb_fraction=0, tie_word_embeddings=False, fused_dropout_add_ln=False,
        fused_bias_fc=False, patch_embed_bias=bool(getattr(cfg, "patch_embed_bias", True)), use_flash_attn=False, qkv_proj_bias=True, mlp_fc1_bias=True,
        mlp_fc2_bias=True, use_rms_norm=False, causal=False, hidden_features_scaling_factor=1.0, mask_token=False,
        learned_pos_embedding=False, patch_dropout=patch_dropout, sinusoidal_pos_embedding=False)
    m = vit.ViTModel(c).float()
    sd = vit_ref.random_state_dict(cfg, seed)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    m.eval()
    g = torch.Generator().manual_seed(seed + 1)
    pixels = torch.randn(5, cfg.num_channels, cfg.img_size, cfg.img_size, generator=g)
    out = {}
    if patch_dropout > 0:
        m.train()
        rng_seed = seed + 3
        P = (cfg.img_size // cfg.patch_size) ** 2
        K = max(1, int(P * (1 - patch_dropout)))
        torch.manual_seed(rng_seed)
        out["keep"] = torch.randn(5, P).topk(K, dim=-1).indices.numpy()   # (what PatchDropout.forward is about to draw)
        out["rng_seed"] = np.array(rng_seed)
        torch.manual_seed(rng_seed)
    hid = m(pixels).last_hidden_state
    if patch_dropout > 0:
        assert hid.shape[1] == 1 + out["keep"].shape[1], hid.shape
    for pooling in ("cls", "mean"):
        m.zero_grad()
        # BiEncoder pooling restated (modeling_biencoder.py:44-49,79-90,317) on the REFERENCE hidden states
        e = hid[:, 0] if pooling == "cls" else hid.mean(1)
        emb = torch.nn.functional.normalize(e, dim=-1)
        probe = torch.randn(emb.shape, generator=torch.Generator().manual_seed(seed + 2))
        (emb * probe).sum().backward(retain_graph=True)
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        out[f"{pooling}/embedding"] = emb.detach().numpy()
        out[f"{pooling}/probe"] = probe.numpy()
        for k, gten in grads.items():
            out[f"{pooling}/gnorm/" + k] = np.array(float(gten.norm()))
        out[f"{pooling}/g/embeddings.cls_token"] = grads["embeddings.cls_token"].numpy()
        out[f"{pooling}/g/embeddings.pos_embed"] = grads["embeddings.pos_embed"].numpy()
        out[f"{pooling}/g/embeddings.proj.weight[:16,:16]"] = grads["embeddings.proj.weight"][:16, :16].numpy()
        out[f"{pooling}/g/layers.0.attn.Wqkv.weight[:16,:16]"] = grads["layers.0.attn.Wqkv.weight"][:16, :16].numpy()
        out[f"{pooling}/g/layers.1.mlp.fc2.weight[:16,:16]"] = grads["layers.1.mlp.fc2.weight"][:16, :16].numpy()
        out[f"{pooling}/g/ln_f.weight"] = grads["ln_f.weight"].numpy()
    np.savez_compressed(GOLD / f"{name}.npz", seed=seed, pixels=pixels.numpy(), hidden=hid.detach().numpy(),
                        weight_checksum=checksum(sd), **out, **{"cfg/" + k: np.array(v) for k, v in cfgd.items()})
    print(name, "hidden", tuple(hid.shape))


def gen_map_pool(name, seed):
    """tests/golden/map_pool_tiny.npz: the reference's own MultiHeadAttentionPooling (modeling_biencoder.py:93-156) on CPU
    in fp32, applied to random (B, S, d) hidden states: output, gradient of the hidden states and of every parameter."""
    from transformers import GPT2Config

    from oracle import map_pool_ref

    bi = ref_import.load_biencoder()
    d, H, inner, eps = 256, 4, 512, 1e-6
    c = GPT2Config(n_embd=d, n_head=H, n_inner=inner, activation_function="gelu", layer_norm_epsilon=eps, attn_pdrop=0.0,
                   use_flash_attn=True, fused_bias_fc=False, qkv_proj_bias=True, mlp_fc1_bias=True, mlp_fc2_bias=True,
                   use_rms_norm=False, causal=False)
    m = bi.MultiHeadAttentionPooling(c).float()
    sd = map_pool_ref.random_state_dict(d, inner, seed)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    g = torch.Generator().manual_seed(seed + 1)
    hidden = torch.randn(5, 17, d, generator=g).requires_grad_()
    probe = torch.randn(5, d, generator=g)
    out = m(hidden, None, None)          # attention_mask=None: the image-tower branch (the masked one cannot run, see map_pooling.py)
    (out * probe).sum().backward()
    rec = {"hidden": hidden.detach().numpy(), "probe": probe.numpy(), "out": out.detach().numpy(),
           "g/hidden": hidden.grad.numpy(), "n_head": np.array(H), "eps": np.array(eps), "seed": np.array(seed)}
    # (weights are map_pool_ref.random_state_dict(d, inner, seed): not stored; matrices keep their norm and a corner)
    for k, p_ in m.named_parameters():
        gk = p_.grad.detach()
        rec["gnorm/" + k] = np.array(float(gk.norm()))
        rec["g/" + k] = (gk[:16, :16] if gk.ndim == 2 else gk).numpy()
    rec["weight_checksum"] = checksum(sd)
    np.savez_compressed(GOLD / f"{name}.npz", d=np.array(d), inner=np.array(inner), **rec)
    print(name, "out", tuple(out.shape), "params", sorted(k for k, _ in m.named_parameters()))


def hf_bert_state_dict(L=2, d=8, inter=16, vocab=10, types=2, pos=12, seed=11, gamma_beta=False, roberta=False):
    """A BertForPreTraining-shaped state dict with HF key names (random values): the input of the remap goldens."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    ln_w, ln_b = ("gamma", "beta") if gamma_beta else ("weight", "bias")
    top = "roberta" if roberta else "bert"
    sd = OrderedDict()
    sd[f"{top}.embeddings.position_ids"] = torch.arange(pos).unsqueeze(0)
    sd[f"{top}.embeddings.word_embeddings.weight"] = r(vocab, d)
    sd[f"{top}.embeddings.position_embeddings.weight"] = r(pos, d)
    sd[f"{top}.embeddings.token_type_embeddings.weight"] = r(types, d)
    sd[f"{top}.embeddings.LayerNorm.{ln_w}"], sd[f"{top}.embeddings.LayerNorm.{ln_b}"] = r(d), r(d)
    for l in range(L):
        p = f"{top}.encoder.layer.{l}."
        for n in ("query", "key", "value"):
            sd[p + f"attention.self.{n}.weight"], sd[p + f"attention.self.{n}.bias"] = r(d, d), r(d)
        sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"] = r(d, d), r(d)
        sd[p + f"attention.output.LayerNorm.{ln_w}"], sd[p + f"attention.output.LayerNorm.{ln_b}"] = r(d), r(d)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = r(inter, d), r(inter)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = r(d, inter), r(d)
        sd[p + f"output.LayerNorm.{ln_w}"], sd[p + f"output.LayerNorm.{ln_b}"] = r(d), r(d)
    sd[f"{top}.pooler.dense.weight"], sd[f"{top}.pooler.dense.bias"] = r(d, d), r(d)
    sd["cls.predictions.bias"] = r(vocab)
    sd["cls.predictions.transform.dense.weight"], sd["cls.predictions.transform.dense.bias"] = r(d, d), r(d)
    sd[f"cls.predictions.transform.LayerNorm.{ln_w}"], sd[f"cls.predictions.transform.LayerNorm.{ln_b}"] = r(d), r(d)
    sd["cls.predictions.decoder.weight"] = sd[f"{top}.embeddings.word_embeddings.weight"].clone()
    sd["cls.seq_relationship.weight"], sd["cls.seq_relationship.bias"] = r(2, d), r(2)
    return sd


HF_REMAP_CASES = {
    # name: (state-dict kwargs, config extras, remap kwargs)
    "plain": (dict(), dict(), dict()),
    "biencoder": (dict(gamma_beta=True), dict(), dict(remove_bert=True, remove_cls_weights=True, add_pooling_layer=False)),
    "padded": (dict(), dict(vocab_size=16, pad_vocab_size_multiple=8, orig_vocab_size=10), dict(add_pooling_layer=True)),
    "roberta_subset_rotary": (dict(roberta=True), dict(last_layer_subset=True, rotary_emb_fraction=1.0), dict()),
}


def gen_hf_remap():
    """tests/golden/hf_remap.npz: outputs of the reference's remap_bert_state_dict / inv_remap_state_dict
    (sc/models/encoder/bert.py:75-366) on small HF-named state dicts; keys are stored in order."""
    from transformers import BertConfig

    rb = ref_import.load_bert_remap()
    out = {}
    for case, (sdkw, cfgkw, kw) in HF_REMAP_CASES.items():
        cfg = BertConfig(vocab_size=10, hidden_size=8, num_hidden_layers=2, num_attention_heads=2, intermediate_size=16,
                         max_position_embeddings=12, type_vocab_size=2)
        for k, v in cfgkw.items():
            setattr(cfg, k, v)
        fwd = rb.remap_bert_state_dict(hf_bert_state_dict(**sdkw), cfg, **kw)
        out[f"{case}/fwd_keys"] = np.array(list(fwd.keys()))
        for k, v in fwd.items():
            out[f"{case}/fwd/{k}"] = v.numpy()
        if not kw.get("remove_bert") and not kw.get("remove_cls_weights"):
            inv = rb.inv_remap_state_dict(OrderedDict((k, v.clone()) for k, v in fwd.items()), cfg)
            out[f"{case}/inv_keys"] = np.array(list(inv.keys()))
            for k, v in inv.items():
                out[f"{case}/inv/{k}"] = v.numpy()
    np.savez_compressed(GOLD / "hf_remap.npz", **out)
    print("hf_remap", {c: len(out[f"{c}/fwd_keys"]) for c in HF_REMAP_CASES})


def make_mlm_inputs(cfgd, B, S, seed, p=0.3):
    ids, mask, lens = make_inputs(cfgd, B, S, seed)
    g = torch.Generator().manual_seed(seed + 7)
    target = (torch.rand(B, S, generator=g) < p) & mask.bool()
    target[0, 1] = True
    labels = torch.where(target, ids, torch.full_like(ids, -100))
    masked = torch.where(target & (torch.rand(B, S, generator=g) < 0.8), torch.full_like(ids, 4), ids)
    return masked, mask, lens, labels


def gen_mlm(name, cfgd, seed):
    """tests/golden/<name>.npz: the reference's eager NomicBertForPreTraining (modeling_hf_nomic_bert.py:1704-1765),
    fp32 CPU: loss, logits of the target positions, gradient norms of every parameter (tied embedding included)."""
    from oracle import mlm_ref

    _, rcfg, rmod = ref_import.load()
    cfg = cfg_ns(cfgd)
    trunk = encoder_ref.random_state_dict(cfg, seed)
    head = mlm_ref.random_head_state_dict(cfg, seed + 100)
    c = ref_model(cfgd, trunk).config
    m = rmod.NomicBertForPreTraining(c)
    sd = {f"bert.{k}": v for k, v in trunk.items()}
    sd.update(head)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("inv_freq" in k or "norm_factor" in k or k == "cls.predictions.decoder.weight" for k in missing), missing
    m.tie_weights()
    m.eval()
    ids, mask, lens, labels = make_mlm_inputs(cfgd, 6, 32, seed + 1)
    out = m(ids, attention_mask=mask, labels=labels)
    out.loss.backward()
    keep = {}
    for k, p_ in m.named_parameters():
        if p_.grad is not None:
            keep["gnorm/" + k] = np.array(float(p_.grad.norm()))
    tgt = labels.flatten() >= 0
    keep["g/cls.predictions.transform.layer_norm.weight"] = m.cls.predictions.transform.layer_norm.weight.grad.numpy()
    keep["g/word_rows"] = m.bert.embeddings.word_embeddings.weight.grad[labels.flatten()[tgt][:8]].numpy()
    np.savez_compressed(GOLD / f"{name}.npz", seed=seed, input_ids=ids.numpy(), attention_mask=mask.numpy(),
                        lens=lens.numpy(), labels=labels.numpy(), loss=np.array(float(out.loss)),
                        target_logits=out.logits.detach().flatten(0, 1)[tgt].numpy().astype(np.float16),
                        weight_checksum=checksum({**trunk, **head}), **keep,
                        **{"cfg/" + k: np.array(v) for k, v in cfgd.items()})
    print(name, "loss", float(out.loss), "targets", int(tgt.sum()))


class _ForgivingDict(dict):
    """The reference's non-download path deletes `path2stream[path]` entries it never created when a shard is exhausted
    (text_text_loader.py:404) -- a KeyError that ends the epoch.  The golden run gets past it with this dict."""

    def __delitem__(self, k):
        if k in self:
            super().__delitem__(k)


def _rank_loader(rank, world, root, tmp):
    from oracle import data_fixture as df

    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29613", rank=rank, world_size=world)
    tl = ref_import.load_text_loader()
    spec = str(Path(root) / "spec.yaml")
    out = {}
    for case, kw in df.LOADER_CASES.items():
        d = tl.StreamingShardDataset(spec, 8, df.ToyTokenizer(), seed=5, verbose=False, run_name=f"gold_{case}", **kw)
        d.path2stream = _ForgivingDict()
        out[f"{case}/len"] = np.array(len(d))
        n = 0
        for b in d:
            out[f"{case}/{n}/dataset_name"] = np.array(b["dataset_name"])
            for k, v in b.items():
                if torch.is_tensor(v):
                    out[f"{case}/{n}/{k}"] = v.numpy().astype(np.float32 if v.is_floating_point() else np.int16)
            n += 1
        out[f"{case}/n_batches"] = np.array(n)
        processed = json.load(open(d.path))
        out[f"{case}/processed_keys"] = np.array(["/".join(k.split("/")[-2:]) for k in processed])
        out[f"{case}/processed_vals"] = np.array(list(processed.values()))
        dist.barrier()
    np.savez(f"{tmp}/loader{rank}.npz", **out)
    dist.destroy_process_group()


def _rank_local_loader(rank, world, root, tmp):
    import random

    from oracle import data_fixture as df

    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29614", rank=rank, world_size=world)
    tl = ref_import.load_text_loader()
    out = {}
    for case, (variant, kw) in df.LOCAL_CASES.items():
        spec = str(Path(root) / f"spec_local_{variant}_s3.yaml")
        random.seed(1234 + rank)  # LocalShardDataset.__getitem__ samples negatives from the global RNG (:757)
        dl = tl.get_local_dataloader(spec, 4, df.ToyTokenizer(), seed=3, **kw)
        n = 0
        for b in dl:
            for k, v in b.items():
                out[f"{case}/{n}/{k}"] = v.numpy().astype(np.int16)
            n += 1
        out[f"{case}/n_batches"] = np.array(n)
        out[f"{case}/len"] = np.array(len(dl.dataset))
        dist.barrier()
    np.savez(f"{tmp}/local{rank}.npz", **out)
    dist.destroy_process_group()


def gen_local_loader():
    """tests/golden/local_loader_w2.npz: every batch of the reference's get_local_dataloader (map-style dataset,
    DistributedSampler, collate_local_ds) on 2 ranks over the toy shards, for the LOCAL_CASES configurations."""
    import shutil
    import tempfile

    from oracle import data_fixture as df

    with tempfile.TemporaryDirectory(prefix="cxlocal_", dir="/tmp") as root, tempfile.TemporaryDirectory() as tmp:
        df.build_dataset(root)
        for variant in {v for v, _ in df.LOCAL_CASES.values()}:
            df.write_local_spec(root, variant, scheme="s3:/")
        try:
            mp.spawn(_rank_local_loader, args=(2, root, tmp), nprocs=2, join=True)
        finally:
            shutil.rmtree("/tmp" + root, ignore_errors=True)  # the reference "downloads" into /tmp/<url path>
        r = [np.load(f"{tmp}/local{i}.npz") for i in range(2)]
        np.savez_compressed(GOLD / "local_loader_w2.npz", **{f"r{i}/{k}": r[i][k] for i in range(2) for k in r[i].files})
    print("local_loader_w2", {c: int(r[0][f"{c}/n_batches"]) for c in df.LOCAL_CASES})


def gen_loader():
    """tests/golden/loader_w2.npz: every batch the reference's StreamingShardDataset yields on 2 ranks for the toy
    shards of oracle/data_fixture.py (global batch 8), for the LOADER_CASES configurations."""
    import tempfile

    from oracle import data_fixture as df

    tl = ref_import.load_text_loader()
    with tempfile.TemporaryDirectory(prefix="cxloader_") as root, tempfile.TemporaryDirectory() as tmp:
        df.build_dataset(root)
        probe = tl.StreamingShardDataset.__new__(tl.StreamingShardDataset)
        df.write_index(root, lambda u: probe.normalize_url([u])[0])
        mp.spawn(_rank_loader, args=(2, root, tmp), nprocs=2, join=True)
        r = [np.load(f"{tmp}/loader{i}.npz") for i in range(2)]
        np.savez_compressed(GOLD / "loader_w2.npz", **{f"r{i}/{k}": r[i][k] for i in range(2) for k in r[i].files})
    print("loader_w2", {c: int(r[0][f"{c}/n_batches"]) for c in df.LOADER_CASES})


class _Scale(torch.nn.Module):
    """Stand-in for LogitScale with a fixed scale (reference passes a module: sc/loss.py:109)."""

    def __init__(self, s):
        super().__init__()
        self.s = s

    def forward(self, x):
        return x * self.s


def gen_clip_loss_single():
    ref_loss, _, _ = ref_import.load()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1)
    out = {}
    g = torch.Generator().manual_seed(7)
    for tag, N, M, bid in (("sq", 8, 8, False), ("neg", 8, 24, False), ("bi", 8, 8, True), ("big", 96, 288, False)):
        q = torch.nn.functional.normalize(torch.randn(N, 64, generator=g), dim=-1).requires_grad_()
        d = torch.nn.functional.normalize(torch.randn(M, 64, generator=g), dim=-1).requires_grad_()
        loss = ref_loss.clip_loss(q, d, _Scale(50.0), bidirectional=bid)
        loss.backward()
        out.update({f"{tag}/q": q.detach().numpy(), f"{tag}/d": d.detach().numpy(), f"{tag}/loss": loss.detach().numpy(),
                    f"{tag}/dq": q.grad.numpy(), f"{tag}/dd": d.grad.numpy()})
    # KAT of the reference's own unit test (tests/test_loss.py:5-17)
    q = torch.tensor([[1.0, 2], [2, 3], [3, 4]])
    d = torch.tensor([[1.0, 2], [3, 4], [2, 3]])
    qn, dn = torch.nn.functional.normalize(q, dim=-1), torch.nn.functional.normalize(d, dim=-1)
    out["kat/q"], out["kat/d"] = qn.numpy(), dn.numpy()
    out["kat/loss"] = ref_loss.clip_loss(qn, dn, _Scale(1.0)).numpy()
    np.savez_compressed(GOLD / "clip_loss_w1.npz", **out)
    dist.destroy_process_group()
    print("clip_loss_w1 done")


def gen_matryoshka_step():
    """tests/golden/matryoshka_step.npz: the reference trainer's direct step (sc/trainers/text_text.py:324-378 `_forward_step`:
    both sides through the model with normalize = False, gather_with_grad, one clip_loss per Matryoshka prefix on re-normalised
    prefixes, weighted sum; hard negatives folded into the document side) -- the REFERENCE'S OWN FUNCTION, compiled from its
    source file (the module itself imports deepspeed / sentence_transformers / megablocks, absent here, so the FunctionDef is
    lifted out of the file with `ast` and executed against the reference's own clip_loss / gather_with_grad).  The model is a
    stand-in that returns fixed embeddings: what is pinned is the loss composition of SURVEY row a20, not the encoder."""
    import ast
    from types import SimpleNamespace

    ref_loss, _, _ = ref_import.load()
    ref_dist = __import__("importlib").import_module("contrastors.distributed")
    src = (ref_import.REF_ROOT / "trainers" / "text_text.py").read_text()
    fn = next(n for c in ast.parse(src).body if isinstance(c, ast.ClassDef) and c.name == "TextTextTrainer"
              for n in c.body if isinstance(n, ast.FunctionDef) and n.name == "_forward_step")
    ns = {"F": torch.nn.functional, "torch": torch, "gather_with_grad": ref_dist.gather_with_grad, "clip_loss": ref_loss.clip_loss,
          "moe": SimpleNamespace(clear_load_balancing_loss=lambda: None), "calculate_auxiliary_loss": None}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), str(ref_import.REF_ROOT / "trainers" / "text_text.py"), "exec"), ns)
    forward_step = ns["_forward_step"]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29614", rank=0, world_size=1)
    out = {}
    g = torch.Generator().manual_seed(31)
    # (query and document counts are multiples of 4: the product's fused InfoNCE backward contracts over them)
    for tag, nq, negs, dims, weights in (("m4", 8, 7, [64, 32, 16, 8], [1.0, 1.0, 1.0, 1.0]), ("w3", 4, 2, [48, 24, 8], [1.0, 0.5, 0.25]),
                                         ("plain", 8, 7, None, None)):
        q = (torch.randn(nq, 64, generator=g) * 1.5).requires_grad_()
        d = (torch.randn(nq * (1 + negs), 64, generator=g) * 1.5).requires_grad_()

        class _Model:   # BiEncoder's output contract; `normalize` as the trainer passes it
            device = torch.device("cpu")

            def __call__(self, input_ids, attention_mask=None, normalize=True):
                e = q if input_ids.shape[0] == nq else d
                return {"embedding": torch.nn.functional.normalize(e, dim=-1) if normalize else e, "router_loss": None}

        trainer = SimpleNamespace(config=SimpleNamespace(model_args=SimpleNamespace(num_experts=0),
                                                         train_args=SimpleNamespace(router_aux_loss_coef=0.0, wandb=False)),
                                  tracker=None)
        batch = {"dataset_name": "golden", "query_input_ids": torch.zeros(nq, 4, dtype=torch.long),
                 "query_attention_mask": torch.ones(nq, 4, dtype=torch.long),
                 "document_input_ids": torch.zeros(nq * (1 + negs), 4, dtype=torch.long),
                 "document_attention_mask": torch.ones(nq * (1 + negs), 4, dtype=torch.long)}
        res = forward_step(trainer, _Model(), batch, _Scale(50.0), matryoshka_dims=dims, matroyshka_loss_weights=weights)
        loss = res["loss"] if isinstance(res, dict) else res
        loss.backward()
        out.update({f"{tag}/q": q.detach().numpy(), f"{tag}/d": d.detach().numpy(), f"{tag}/loss": loss.detach().numpy(),
                    f"{tag}/dq": q.grad.numpy(), f"{tag}/dd": d.grad.numpy(),
                    f"{tag}/dims": np.array(dims if dims else [], dtype=np.int64),
                    f"{tag}/weights": np.array(weights if weights else [], dtype=np.float64)})
    np.savez_compressed(GOLD / "matryoshka_step.npz", **out)
    dist.destroy_process_group()
    print("matryoshka_step done")


def _rank_clip(rank, world, tmp):
    ref_loss, _, _ = ref_import.load()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29612", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    q = torch.nn.functional.normalize(torch.randn(4, 64, generator=g), dim=-1).requires_grad_()
    d = torch.nn.functional.normalize(torch.randn(12, 64, generator=g), dim=-1).requires_grad_()  # 1 pos + 2 neg
    loss = ref_loss.clip_loss(q, d, _Scale(50.0), gather_enabled=True)
    loss.backward()
    np.savez(f"{tmp}/r{rank}.npz", q=q.detach().numpy(), d=d.detach().numpy(), loss=loss.detach().numpy(),
             dq=q.grad.numpy(), dd=d.grad.numpy())
    dist.destroy_process_group()


class _Tower(torch.nn.Module):
    """Reference eager trunk + BiEncoder mean-pool/normalise restatement, shaped like BiEncoder's output dict."""

    def __init__(self, trunk):
        super().__init__()
        self.trunk = trunk

    def forward(self, input_ids, attention_mask=None, **kw):
        h = self.trunk(input_ids, attention_mask=attention_mask).last_hidden_state
        s = (h * attention_mask.unsqueeze(-1).float()).sum(1) / attention_mask.sum(1, keepdim=True).float()
        return {"embedding": torch.nn.functional.normalize(s, dim=-1)}


def _rank_gradcache(rank, world, tmp):
    warnings.filterwarnings("ignore")
    ref_loss, _, _ = ref_import.load()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29613", rank=rank, world_size=world)
    cfg = cfg_ns(TINY_NOMIC)
    sd = encoder_ref.random_state_dict(cfg, 11)
    tower = _Tower(ref_model(TINY_NOMIC, sd)).train()
    ddp = torch.nn.parallel.DistributedDataParallel(tower, broadcast_buffers=False)
    ddp.device = torch.device("cpu")
    qi, qm, _ = make_inputs(TINY_NOMIC, 4, 16, 300 + rank)
    di, dm, _ = make_inputs(TINY_NOMIC, 4, 32, 400 + rank)
    loss = ref_loss.grad_cache_loss(ddp, {"input_ids": qi, "attention_mask": qm}, ddp,
                                    {"input_ids": di, "attention_mask": dm}, chunk_size=2,
                                    logit_scale=_Scale(20.0))
    grads = {k: p.grad.detach().clone() for k, p in tower.trunk.named_parameters() if p.grad is not None}
    keep = {"gnorm/" + k: np.array(float(v.norm())) for k, v in grads.items()}
    keep["g/emb_ln.weight"] = grads["emb_ln.weight"].numpy()
    keep["g/encoder.layers.1.attn.out_proj.weight[:16,:16]"] = grads["encoder.layers.1.attn.out_proj.weight"][:16, :16].numpy()
    np.savez(f"{tmp}/gc{rank}.npz", q_ids=qi.numpy(), q_mask=qm.numpy(), d_ids=di.numpy(), d_mask=dm.numpy(),
             loss=loss.detach().numpy(), **keep)
    dist.destroy_process_group()


def gen_multirank():
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_rank_clip, args=(2, tmp), nprocs=2, join=True)
        r = [np.load(f"{tmp}/r{i}.npz") for i in range(2)]
        np.savez_compressed(GOLD / "clip_loss_w2.npz", **{f"r{i}/{k}": r[i][k] for i in range(2) for k in r[i].files})
        mp.spawn(_rank_gradcache, args=(2, tmp), nprocs=2, join=True)
        r = [np.load(f"{tmp}/gc{i}.npz") for i in range(2)]
        np.savez_compressed(GOLD / "grad_cache_w2.npz", seed=11,
                            **{f"r{i}/{k}": r[i][k] for i in range(2) for k in r[i].files})
    print("multi-rank fixtures done")


if __name__ == "__main__":
    warnings.filterwarnings("ignore")
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    GOLD.mkdir(parents=True, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "vit":  # regenerate only the ViT fixtures
        gen_vit("vit_tiny", TINY_VIT, 5)
        gen_vit("vit_clip_tiny", TINY_VIT_CLIP, 6)
        gen_vit("vit_patchdrop_tiny", TINY_VIT, 7, patch_dropout=0.5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "map_pool":
        gen_map_pool("map_pool_tiny", 9)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mlm":
        gen_mlm("mlm_nomic_tiny", TINY_NOMIC, 21)
        gen_mlm("mlm_bert_tiny", TINY_BERT, 22)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "loader":
        gen_loader()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "local_loader":
        gen_local_loader()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "matryoshka":
        gen_matryoshka_step()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "hf_remap":
        gen_hf_remap()
        sys.exit(0)
    gen_encoder("encoder_nomic_tiny", TINY_NOMIC, 1)
    gen_encoder("encoder_bert_tiny", TINY_BERT, 2)
    # Dynamic-NTK rotary (modeling_hf_nomic_bert.py:1215-1235): sequences (32) longer than max_trained_positions (16)
    gen_encoder("encoder_nomic_ntk_tiny", dict(TINY_NOMIC, rotary_scaling_factor=2.0, max_trained_positions=16), 5)
    gen_clip_loss_single()
    gen_matryoshka_step()
    gen_multirank()
    gen_vit("vit_tiny", TINY_VIT, 5)
    gen_vit("vit_clip_tiny", TINY_VIT_CLIP, 6)
    gen_vit("vit_patchdrop_tiny", TINY_VIT, 7, patch_dropout=0.5)
    gen_map_pool("map_pool_tiny", 9)
    gen_hf_remap()
    gen_mlm("mlm_nomic_tiny", TINY_NOMIC, 21)
    gen_mlm("mlm_bert_tiny", TINY_BERT, 22)
    gen_loader()
    gen_local_loader()
