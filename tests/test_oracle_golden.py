"""The oracle (oracle/*.py, a CPU restatement) is pinned against fixtures produced by the REFERENCE's own code
(oracle/make_golden.py imports /root/reference ... loss.py and modeling_hf_nomic_bert.py).  CPU only."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import encoder_ref, infonce_ref, vit_ref


def _cfg(g):
    d = {}
    for k in g.files:
        if k.startswith("cfg/"):
            v = g[k]
            d[k[4:]] = v.item() if v.shape == () else v
    return SimpleNamespace(**{k: (str(v) if isinstance(v, (str, np.str_)) else v) for k, v in d.items()})


@pytest.mark.parametrize("name", ["encoder_nomic_tiny", "encoder_bert_tiny", "encoder_nomic_ntk_tiny"])
def test_encoder_restatement_matches_reference(gold, name):
    g = gold(name)
    cfg = _cfg(g)
    sd = encoder_ref.random_state_dict(cfg, int(g["seed"]))
    cs = np.array([float(sum(v.double().sum() for v in sd.values())),
                   float(sum((v.double() ** 2).sum() for v in sd.values()))])
    np.testing.assert_allclose(cs, g["weight_checksum"], rtol=1e-12)
    for v in sd.values():
        v.requires_grad_(True)
    ids = torch.from_numpy(g["input_ids"])
    mask = torch.from_numpy(g["attention_mask"])
    hid = encoder_ref.encoder_hidden_states(sd, cfg, ids, mask)
    np.testing.assert_allclose((hid * mask.unsqueeze(-1)).detach().numpy(), g["hidden"], atol=2e-5, rtol=1e-4)
    emb = encoder_ref.biencoder_embedding(sd, cfg, ids, mask)
    np.testing.assert_allclose(emb.detach().numpy(), g["embedding"], atol=2e-6)
    (emb * torch.from_numpy(g["probe"])).sum().backward()
    for k in g.files:
        if k.startswith("gnorm/"):
            name_ = k[6:]
            assert abs(float(sd[name_].grad.norm()) - float(g[k])) <= 1e-4 * max(1.0, float(g[k])), name_
    np.testing.assert_allclose(sd["emb_ln.weight"].grad.numpy(), g["g/emb_ln.weight"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(sd["encoder.layers.0.attn.Wqkv.weight"].grad[:16, :16].numpy(),
                               g["g/encoder.layers.0.attn.Wqkv.weight[:16,:16]"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(sd["embeddings.word_embeddings.weight"].grad[ids[0, :8]].numpy(),
                               g["g/embeddings.word_embeddings.weight[rows]"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("tag,bid", [("sq", False), ("neg", False), ("bi", True), ("big", False)])
def test_clip_loss_restatement_w1(gold, tag, bid):
    g = gold("clip_loss_w1")
    q = torch.from_numpy(g[f"{tag}/q"]).requires_grad_()
    d = torch.from_numpy(g[f"{tag}/d"]).requires_grad_()
    loss = infonce_ref.clip_loss_ref(q, d, 50.0, bidirectional=bid)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{tag}/loss"], rtol=1e-5)
    np.testing.assert_allclose(q.grad.numpy(), g[f"{tag}/dq"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(d.grad.numpy(), g[f"{tag}/dd"], atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("tag", ["m4", "w3", "plain"])
def test_matryoshka_step_restatement(gold, tag):
    """The reference trainer's own `_forward_step` (sc/trainers/text_text.py:324-378, lifted from its source file by
    oracle/make_golden.py) on fixed embeddings: 1 positive + 7 (or 2) hard negatives per query, Matryoshka prefixes with unit
    and non-unit weights, and the plain (normalised, single-loss) case."""
    g = gold("matryoshka_step")
    q = torch.from_numpy(g[f"{tag}/q"]).requires_grad_()
    d = torch.from_numpy(g[f"{tag}/d"]).requires_grad_()
    loss = infonce_ref.matryoshka_step_loss_ref(q, d, 50.0, [int(x) for x in g[f"{tag}/dims"]], list(g[f"{tag}/weights"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{tag}/loss"], rtol=1e-5)
    np.testing.assert_allclose(q.grad.numpy(), g[f"{tag}/dq"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(d.grad.numpy(), g[f"{tag}/dd"], atol=1e-6, rtol=1e-4)


def test_reference_kat_toy_infonce(gold):
    """tests/test_loss.py:5-17 of the reference: clip_loss == -mean log softmax on the diagonal labels."""
    g = gold("clip_loss_w1")
    q, d = torch.from_numpy(g["kat/q"]), torch.from_numpy(g["kat/d"])
    naive = -torch.log_softmax(q @ d.T, dim=1)[torch.arange(3), torch.arange(3)].mean()
    np.testing.assert_allclose(naive.item(), g["kat/loss"], rtol=1e-6)
    np.testing.assert_allclose(infonce_ref.clip_loss_ref(q, d, 1.0).item(), g["kat/loss"], rtol=1e-6)
    lse, rows = infonce_ref.infonce_rows_np(g["kat/q"], g["kat/d"], np.arange(3), 1.0)
    np.testing.assert_allclose(rows.mean(), g["kat/loss"], rtol=1e-6)


def test_labels_bit_exact():
    # sc/loss.py:108-117 with hard negatives: stride 1+N_neg, rank offset
    np.testing.assert_array_equal(infonce_ref.labels_for(4, 24, 1, 2), np.array([12, 15, 18, 21], dtype=np.int64))
    np.testing.assert_array_equal(infonce_ref.labels_for(3, 3, 0, 1), np.arange(3))


def test_clip_loss_restatement_two_ranks(gold):
    g = gold("clip_loss_w2")
    qs = [torch.from_numpy(g[f"r{r}/q"]).requires_grad_() for r in range(2)]
    ds = [torch.from_numpy(g[f"r{r}/d"]).requires_grad_() for r in range(2)]
    losses = infonce_ref.multi_rank_clip_loss_ref(qs, ds, 50.0)
    sum(losses).backward()
    for r in range(2):
        np.testing.assert_allclose(losses[r].item(), g[f"r{r}/loss"], rtol=1e-5)
        np.testing.assert_allclose(qs[r].grad.numpy(), g[f"r{r}/dq"], atol=1e-6, rtol=1e-4)
        np.testing.assert_allclose(ds[r].grad.numpy(), g[f"r{r}/dd"], atol=1e-6, rtol=1e-4)


def test_gradcache_ddp_two_ranks(gold):
    """Reference grad_cache_loss under 2-rank gloo DDP == full-batch loss, gradients averaged over ranks."""
    from oracle.make_golden import TINY_NOMIC

    g = gold("grad_cache_w2")
    cfg = SimpleNamespace(**TINY_NOMIC)
    sd = encoder_ref.random_state_dict(cfg, int(g["seed"]))
    for v in sd.values():
        v.requires_grad_(True)
    qs, ds = [], []
    for r in range(2):
        qs.append(encoder_ref.biencoder_embedding(sd, cfg, torch.from_numpy(g[f"r{r}/q_ids"]),
                                                  torch.from_numpy(g[f"r{r}/q_mask"])))
        ds.append(encoder_ref.biencoder_embedding(sd, cfg, torch.from_numpy(g[f"r{r}/d_ids"]),
                                                  torch.from_numpy(g[f"r{r}/d_mask"])))
    losses = infonce_ref.multi_rank_clip_loss_ref(qs, ds, 20.0)
    (sum(losses) / 2).backward()  # DDP averages parameter gradients over the 2 ranks
    for r in range(2):
        np.testing.assert_allclose(losses[r].item(), g[f"r{r}/loss"], rtol=2e-5)
    np.testing.assert_allclose(sd["emb_ln.weight"].grad.numpy(), g["r0/g/emb_ln.weight"], atol=3e-5, rtol=1e-3)
    np.testing.assert_allclose(sd["encoder.layers.1.attn.out_proj.weight"].grad[:16, :16].numpy(),
                               g["r0/g/encoder.layers.1.attn.out_proj.weight[:16,:16]"], atol=3e-5, rtol=1e-3)
    for k in g.files:
        if k.startswith("r0/gnorm/"):
            n = k[len("r0/gnorm/"):]
            assert abs(float(sd[n].grad.norm()) - float(g[k])) <= 2e-3 * max(1e-3, float(g[k])), n


@pytest.mark.parametrize("name", ["vit_tiny", "vit_clip_tiny", "vit_patchdrop_tiny"])
def test_vit_restatement_matches_reference(gold, name):
    """oracle/vit_ref.py vs the reference's own ViTModel python (sc/models/vit/vit.py) on CPU fp32: hidden states,
    pooled embeddings for both poolings, every parameter-gradient norm and five gradient slices.  vit_clip_tiny = the
    OpenAI-CLIP flavour (sc/models/vit/clip.py:14-58): quick_gelu, pre-LayerNorm, no patch-embedding bias."""
    g = gold(name)
    cfg = _cfg(g)
    sd = vit_ref.random_state_dict(cfg, int(g["seed"]))
    cs = np.array([float(sum(v.double().sum() for v in sd.values())),
                   float(sum((v.double() ** 2).sum() for v in sd.values()))])
    np.testing.assert_allclose(cs, g["weight_checksum"], rtol=1e-12)
    for v in sd.values():
        v.requires_grad_(True)
    pix = torch.from_numpy(g["pixels"])
    # vit_patchdrop_tiny: the reference model in training mode with PatchDropout 0.5; `keep` = what its CPU draw selected
    keep = torch.from_numpy(g["keep"]) if "keep" in g.files else None
    np.testing.assert_allclose(vit_ref.vit_hidden(sd, cfg, pix, keep).detach().numpy(), g["hidden"], atol=2e-5, rtol=1e-4)
    for pooling in ("cls", "mean"):
        for v in sd.values():
            v.grad = None
        emb = vit_ref.vit_embedding(sd, cfg, pix, pooling, keep=keep)
        np.testing.assert_allclose(emb.detach().numpy(), g[f"{pooling}/embedding"], atol=2e-6)
        (emb * torch.from_numpy(g[f"{pooling}/probe"])).sum().backward()
        for k in g.files:
            if k.startswith(f"{pooling}/gnorm/"):
                n = k[len(pooling) + 7:]
                assert abs(float(sd[n].grad.norm()) - float(g[k])) <= 1e-4 * max(1.0, float(g[k])), n
        for n, sl in (("embeddings.cls_token", None), ("embeddings.pos_embed", None), ("ln_f.weight", None),
                      ("embeddings.proj.weight", 16), ("layers.0.attn.Wqkv.weight", 16), ("layers.1.mlp.fc2.weight", 16)):
            got = sd[n].grad if sl is None else sd[n].grad[:sl, :sl]
            key = f"{pooling}/g/{n}" + ("" if sl is None else "[:16,:16]")
            np.testing.assert_allclose(got.numpy(), g[key], atol=2e-5, rtol=1e-4)


def test_map_pooling_restatement_matches_reference(gold):
    """oracle/map_pool_ref.py vs the reference's own MultiHeadAttentionPooling (modeling_biencoder.py:93-156; `pooling: map`
    of the vision recipes) on CPU fp32: output, gradient of the hidden states, every parameter-gradient norm and slice."""
    from oracle import map_pool_ref

    g = gold("map_pool_tiny")
    sd = map_pool_ref.random_state_dict(int(g["d"]), int(g["inner"]), int(g["seed"]))
    cs = np.array([float(sum(v.double().sum() for v in sd.values())),
                   float(sum((v.double() ** 2).sum() for v in sd.values()))])
    np.testing.assert_allclose(cs, g["weight_checksum"], rtol=1e-12)
    for v in sd.values():
        v.requires_grad_(True)
    hidden = torch.from_numpy(g["hidden"]).requires_grad_()
    out = map_pool_ref.map_pool(sd, hidden, int(g["n_head"]), float(g["eps"]))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], atol=2e-6, rtol=1e-5)
    (out * torch.from_numpy(g["probe"])).sum().backward()
    np.testing.assert_allclose(hidden.grad.numpy(), g["g/hidden"], atol=2e-6, rtol=1e-4)
    for k in g.files:
        if k.startswith("gnorm/"):
            n = k[6:]
            assert abs(float(sd[n].grad.norm()) - float(g[k])) <= 1e-4 * max(1e-3, float(g[k])), n
            got = sd[n].grad[:16, :16] if sd[n].grad.ndim == 2 else sd[n].grad
            np.testing.assert_allclose(got.numpy(), g["g/" + n], atol=2e-6, rtol=1e-4)


@pytest.mark.parametrize("name", ["mlm_nomic_tiny", "mlm_bert_tiny"])
def test_mlm_restatement_matches_reference(gold, name):
    """oracle/mlm_ref.py vs the reference's eager NomicBertForPreTraining (loss, target logits, every gradient norm)."""
    from oracle import mlm_ref

    g = gold(name)
    cfg = _cfg(g)
    trunk = encoder_ref.random_state_dict(cfg, int(g["seed"]))
    head = mlm_ref.random_head_state_dict(cfg, int(g["seed"]) + 100)
    allp = {**trunk, **head}
    cs = np.array([float(sum(v.double().sum() for v in allp.values())),
                   float(sum((v.double() ** 2).sum() for v in allp.values()))])
    np.testing.assert_allclose(cs, g["weight_checksum"], rtol=1e-12)
    for v in allp.values():
        v.requires_grad_(True)
    ids, mask, labels = (torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels"))
    logits = mlm_ref.mlm_logits(trunk, head, cfg, ids, mask)
    tgt = labels.flatten() >= 0
    np.testing.assert_allclose(logits.detach().flatten(0, 1)[tgt].numpy(), g["target_logits"].astype(np.float32),
                               atol=2e-2, rtol=2e-3)  # fixture stores fp16
    loss = mlm_ref.mlm_loss(trunk, head, cfg, ids, mask, labels)
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-5
    loss.backward()
    for k in g.files:
        if not k.startswith("gnorm/"):
            continue
        n = k[6:]
        if n == "cls.predictions.decoder.weight":
            continue
        p = allp[n[5:]] if n.startswith("bert.") else allp[n]
        assert abs(float(p.grad.norm()) - float(g[k])) <= 1e-4 * max(1.0, float(g[k])), n
    np.testing.assert_allclose(head["cls.predictions.transform.layer_norm.weight"].grad.numpy(),
                               g["g/cls.predictions.transform.layer_norm.weight"], atol=2e-5, rtol=1e-4)
    rows = labels.flatten()[tgt][:8]
    np.testing.assert_allclose(trunk["embeddings.word_embeddings.weight"].grad[rows].numpy(), g["g/word_rows"],
                               atol=2e-5, rtol=1e-4)
