"""Dropout > 0 on the native path (SURVEY.md row a7 + K5 with p > 0): resid_pdrop / embd_pdrop of the text trunk, Philox
masks drawn from torch's device generator, RandContext (sc/rand_state.py:6-22) replay in grad_cache_loss
(sc/loss.py:141-145,156-158).  Masks cannot be compared with the reference's bit for bit (different generators), so:
  * the fused op is checked against torch with the mask the kernel itself used (recovered from its z output);
  * keep rate and rescaling are checked in distribution;
  * replay is checked exactly: same generator state -> identical masks, forward == re-forward, gradients reproducible."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

from contrastors_amd import _C
from contrastors_amd.biencoder import BiEncoder, BiEncoderConfig, LogitScale
from contrastors_amd.loss import grad_cache_loss
from contrastors_amd.nomic_bert import NomicBertConfig
from contrastors_amd.rand_state import RandContext
from tests.gpu_util import L, S, rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_dropout_add_layernorm_kernel_vs_torch_with_the_same_mask(p):
    rows, d = 301, 768
    g = torch.Generator().manual_seed(1)
    x0 = (0.3 * torch.randn(rows, d, generator=g) + 3.0).to(DEV).to(torch.bfloat16)  # bounded away from 0: mask recoverable
    res = torch.randn(rows, d, generator=g).to(DEV).to(torch.bfloat16)
    gam = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    bet = (0.1 * torch.randn(d, generator=g)).to(DEV)
    out, z = torch.empty_like(x0), torch.empty_like(x0)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    seed, off, site = 1234567, 40, 3
    args = (x0.data_ptr(), res.data_ptr(), gam.data_ptr(), bet.data_ptr())
    _C.check(L().cx_dropout_add_layernorm_fwd(*args, out.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, d,
                                              1e-12, p, seed, off, site, S()))
    keep = ((z.float() - res.float()).abs() > 0.5)                   # z = x0 * m / (1-p) + res with |x0| >~ 1
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p)) < 0.01, rate
    xr, rr = x0.float().requires_grad_(), res.float().requires_grad_()
    gr, br = gam.clone().requires_grad_(), bet.clone().requires_grad_()
    zr = xr * keep / (1 - p) + rr
    ref = F.layer_norm(zr, (d,), gr, br, 1e-12)
    assert rel_err(z.float(), zr) < 1e-2 and rel_err(out.float(), ref) < 1e-2      # z and out are stored in bf16
    # same (seed, offset, site) -> same mask; another offset or site -> another mask
    z2, z3, z4 = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
    for zz, (o_, s_) in ((z2, (off, site)), (z3, (off + 4, site)), (z4, (off, site + 1))):
        _C.check(L().cx_dropout_add_layernorm_fwd(*args, out.data_ptr(), zz.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows,
                                                  d, 1e-12, p, seed, o_, s_, S()))
    assert torch.equal(z, z2) and not torch.equal(z, z3) and not torch.equal(z, z4)
    # backward: dz (the residual's gradient) and dx0 = dz * mask / (1 - p)
    _C.check(L().cx_dropout_add_layernorm_fwd(*args, out.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, d,
                                              1e-12, p, seed, off, site, S()))
    do = torch.randn(rows, d, generator=g).to(DEV).to(torch.bfloat16)
    dz, dx0 = torch.empty_like(z), torch.empty_like(z)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    _C.check(L().cx_dropout_add_layernorm_bwd(do.data_ptr(), None, z.data_ptr(), gam.data_ptr(), mean.data_ptr(),
                                              rstd.data_ptr(), dz.data_ptr(), dx0.data_ptr(), dg.data_ptr(), db.data_ptr(), None,
                                              0, rows, d, p, seed, off, site, S()))
    ref.backward(do.float())
    assert rel_err(dz.float(), rr.grad) < 1e-2 and rel_err(dx0.float(), xr.grad) < 1e-2
    assert rel_err(dg, gr.grad) < 1e-2 and rel_err(db, br.grad) < 1e-3
    report("dropout_ln", p=p, keep_rate=rate)


def _tower(p_resid, p_embd, n_layer=2):
    cfg = NomicBertConfig.nomic_bert_2048(vocab_size=1024, n_layer=n_layer, resid_pdrop=p_resid, embd_pdrop=p_embd)
    return BiEncoder(BiEncoderConfig(pooling="mean", trunk_config=cfg), device=DEV, seed=2)


def _batch(n, S_, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(S_ // 2, S_ + 1, (n,), generator=g)
    ids = torch.randint(5, 1024, (n, S_), generator=g)
    mask = (torch.arange(S_)[None] < lens[:, None]).long()
    return {"input_ids": (ids * mask).to(DEV), "attention_mask": mask.to(DEV)}


def test_engine_dropout_train_vs_eval_and_randcontext_replay():
    tower = _tower(0.1, 0.1).train()
    b = _batch(8, 48, 3)
    with torch.no_grad():
        snap = RandContext(b)                       # generator state before the first forward
        e1 = tower(**b)["embedding"].clone()
        e2 = tower(**b)["embedding"].clone()        # the generator advanced: another mask
        with snap:
            e3 = tower(**b)["embedding"].clone()    # replayed state: the first mask again
        e_eval = tower.eval()(**b)["embedding"].clone()
        e_eval2 = tower(**b)["embedding"].clone()
    assert not torch.equal(e1, e2) and torch.equal(e1, e3)
    assert torch.equal(e_eval, e_eval2) and not torch.equal(e_eval, e1)      # eval: no dropout, deterministic
    # dropout is noise of the right size around the eval embedding, not a different function
    assert float((e1 - e_eval).abs().max()) < 0.5 and float(F.cosine_similarity(e1, e_eval).min()) > 0.8


def test_grad_cache_with_dropout_uses_randcontext():
    """Pass 2 must see pass 1's masks (sc/loss.py:141-145,156-158): with the replay the step is reproducible from a given
    generator state and its gradient is the gradient of the loss that was actually computed."""
    scale = LogitScale(SimpleNamespace(logit_scale=20.0, trainable_logit_scale=False)).to(DEV)
    q, d = _batch(16, 32, 5), _batch(16, 32, 6)
    grads, losses = [], []
    for _ in range(2):
        tower = _tower(0.1, 0.1).train()
        torch.manual_seed(77)                       # seeds the device generator too
        tower.trunk.zero_grad()
        losses.append(float(grad_cache_loss(tower, q, tower, d, 4, scale)))
        grads.append(tower.trunk.flat_grad.clone())
    # (type-embedding / emb_ln / bias gradients are reduced with fp32 atomics: summation-order noise only)
    assert losses[0] == losses[1]
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-4 * float(grads[0].abs().max())
    # directional-derivative check of the GradCache gradient under dropout: L(theta + eps g) - L(theta) ~ eps |g|^2, with
    # the SAME masks in both evaluations (generator re-seeded) -- it fails if pass 2 had drawn fresh masks
    tower = _tower(0.1, 0.1).train()
    torch.manual_seed(77)
    tower.trunk.zero_grad()
    l0 = float(grad_cache_loss(tower, q, tower, d, 4, scale))
    gvec = tower.trunk.flat_grad.clone()
    lo, hi = tower.trunk._layout["encoder.layers.0.attn.Wqkv.weight"][0], tower.trunk.n_decay
    direction = torch.zeros_like(gvec)
    direction[lo:hi] = gvec[lo:hi]                  # Linear weights only (bf16 shadows exist for them)
    eps = 2e-2 / float(direction.norm())
    with torch.no_grad():
        tower.trunk.flat_param.add_(direction, alpha=eps)
    tower.trunk.sync_shadows()
    torch.manual_seed(77)
    tower.trunk.zero_grad()
    l1 = float(grad_cache_loss(tower, q, tower, d, 4, scale))
    predicted = eps * float((direction * gvec).sum())
    report("dropout_gradcache", l0=l0, l1=l1, predicted=predicted, actual=l1 - l0)
    assert predicted > 0 and abs((l1 - l0) - predicted) < 0.35 * predicted + 2e-3
