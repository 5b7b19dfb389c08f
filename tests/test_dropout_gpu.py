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
    # the same backward with the column sums of dx0 riding along (round 6: the bias gradient of the Linear whose dropped output this
    # LayerNorm consumed): identical dz / dx0, colsum = the fp32 sum of the bf16 dx0, accumulated INTO the buffer, deterministic
    dz2, dx02 = torch.empty_like(z), torch.empty_like(z)
    dg2, db2 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    cs = torch.full((d,), 0.5, device=DEV)
    ws = torch.empty(3 * d * 256 + 7, device=DEV)
    _C.check(L().cx_dropout_add_layernorm_bwd_colsum(do.data_ptr(), None, z.data_ptr(), gam.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                     dz2.data_ptr(), dx02.data_ptr(), dg2.data_ptr(), db2.data_ptr(), cs.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), rows, d, p, seed, off, site, S()))
    assert torch.equal(dz2, dz) and torch.equal(dx02, dx0)
    assert rel_err(cs, 0.5 + dx0.float().sum(0)) < 1e-5
    assert rel_err(dg2, gr.grad) < 1e-2 and rel_err(db2, br.grad) < 1e-3
    cs_b = torch.full((d,), 0.5, device=DEV)
    _C.check(L().cx_dropout_add_layernorm_bwd_colsum(do.data_ptr(), None, z.data_ptr(), gam.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                     dz2.data_ptr(), dx02.data_ptr(), dg2.data_ptr(), db2.data_ptr(), cs_b.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), rows, d, p, seed, off, site, S()))
    assert torch.equal(cs, cs_b)
    assert L().cx_dropout_add_layernorm_bwd_colsum(do.data_ptr(), None, z.data_ptr(), gam.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                   dz2.data_ptr(), dx02.data_ptr(), dg2.data_ptr(), db2.data_ptr(), cs_b.data_ptr(), None, 0,
                                                   rows, d, p, seed, off, site, S()) == -3   # CX_ERR_ARG: the column sums need the workspace
    report("dropout_ln", p=p, keep_rate=rate)


def _tower(p_resid, p_embd, n_layer=2, attn=0.0):
    cfg = NomicBertConfig.nomic_bert_2048(vocab_size=1024, n_layer=n_layer, resid_pdrop=p_resid, embd_pdrop=p_embd,
                                          attn_pdrop=attn)
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


def test_dropout_streams_of_consecutive_chunks_do_not_alias():
    """the kernels draw site s of a chunk from Philox counter (offset + s); a chunk owns 3 L + 1 sites, so the
    generator has to advance past all of them or chunk k + 1's site s replays chunk k's site s + 4."""
    from contrastors_amd.nomic_bert import NomicBertEngine

    NL = 2
    tower = _tower(0.1, 0.1, n_layer=NL, attn=0.1).train()
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    b = _batch(4, 32, 5)
    o0 = gen.get_offset()
    with torch.no_grad():
        tower(**b)
    o1 = gen.get_offset()
    assert o1 - o0 == NomicBertEngine.dropout_offset_stride(NL) >= 3 * NL + 1 and (o1 - o0) % 4 == 0
    # and at the kernel: the mask of (offset, site 4) == the mask of (offset + 4, site 0) -- which is why 4 was not enough
    # -- while (offset + stride, any site of the next chunk) shares no counter with this chunk's sites
    rows, d = 64, 768
    x0 = torch.randn(rows, d, device=DEV).to(torch.bfloat16)
    g, bb = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)

    def z_of(off, site):
        out, z = torch.empty_like(x0), torch.empty_like(x0)
        mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
        _C.check(L().cx_dropout_add_layernorm_fwd(x0.data_ptr(), 0, g.data_ptr(), bb.data_ptr(), out.data_ptr(), z.data_ptr(),
                                                 mean.data_ptr(), rstd.data_ptr(), rows, d, 1e-5, 0.5, 77, off, site, S()))
        return z
    assert torch.equal(z_of(100, 4), z_of(104, 0))
    stride = NomicBertEngine.dropout_offset_stride(NL)
    cur = [z_of(100, s) for s in range(3 * NL + 1)]
    nxt = [z_of(100 + stride, s) for s in range(3 * NL + 1)]
    assert not any(torch.equal(a, c) for a in cur for c in nxt)


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


@pytest.mark.parametrize("S,lens", [(128, [128, 77, 128]), (320, [320, 200]), (128, [128, 128, 5, 64, 1, 127]), (64, [64, 33]),
                                    (200, [197, 200, 130, 3]), (256, [256, 129, 225])])   # 128 < S <= 256: the single-pass K / V-resident kernels
def test_attention_dropout_matches_torch_with_the_extracted_mask(S, lens):
    """attn_pdrop > 0 (flash_attn_varlen_qkvpacked_func(dropout_p > 0), sc/layers/attention.py:158-182): O = (P * keep /
    (1 - p)) V, dqkv through the same mask; keep(b, h, q, key) read back with the dev library's mask kernel."""
    import math

    import contrastors_amd.flash_attn_api as fa

    H, D, p = 4, 64, 0.2
    B = len(lens)
    g = torch.Generator().manual_seed(12)
    T = sum(lens)
    qkv = (torch.randn(T, 3, H, D, generator=g) * 0.8).to(DEV).bfloat16().requires_grad_()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    torch.manual_seed(31)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    seed, off = gen.initial_seed(), gen.get_offset()
    out = fa.flash_attn_varlen_qkvpacked_func(qkv, cu, max(lens), p, softmax_scale=1.0 / math.sqrt(D))
    go = torch.randn(T, H, D, generator=g).to(DEV).bfloat16()
    out.backward(go)

    keep = torch.empty(B, H, S, S, dtype=torch.uint8, device=DEV)
    _C.check(_C.dev_lib().cx_attn_dropout_keep_mask(keep.data_ptr(), B, H, S, p, seed, off, 0, _C.cur_stream()))
    rate = keep[0, :, : lens[0], : lens[0]].float().mean().item()
    assert abs(rate - (1 - p)) < 0.01, rate
    x = qkv.detach().float().requires_grad_()
    outs = []
    for b, n in enumerate(lens):
        s0 = int(cu[b])
        q, k, v = x[s0:s0 + n, 0], x[s0:s0 + n, 1], x[s0:s0 + n, 2]
        P = torch.softmax(torch.einsum("qhd,khd->hqk", q, k) / math.sqrt(D), dim=-1)
        outs.append(torch.einsum("hqk,khd->qhd", P * keep[b, :, :n, :n].float() / (1 - p), v))
    ref = torch.cat(outs)
    ref.backward(go.float())
    e_o, e_g = rel_err(out.float(), ref), rel_err(qkv.grad.float(), x.grad)
    report("attn_dropout", S=S, keep_rate=rate, e_out=e_o, e_dqkv=e_g)
    assert e_o < 1e-2 and e_g < 2e-2
    # reproducible from the generator state; p = 0 is the plain kernel
    torch.manual_seed(31)
    out2 = fa.flash_attn_varlen_qkvpacked_func(qkv.detach(), cu, max(lens), p, softmax_scale=1.0 / math.sqrt(D))
    assert torch.equal(out2, out.detach())


def test_engine_with_attention_dropout_gradcache_is_reproducible_and_consistent():
    """attn_pdrop through the native trunk: train != eval, the GradCache step is reproducible from the generator state
    (pass 2 sees pass 1's masks), and its gradient is the gradient of the loss that was computed (directional check)."""
    scale = LogitScale(SimpleNamespace(logit_scale=20.0, trainable_logit_scale=False)).to(DEV)
    q, d = _batch(16, 32, 5), _batch(16, 32, 6)
    res = []
    for _ in range(2):
        tower = _tower(0.0, 0.0, attn=0.15).train()
        torch.manual_seed(78)
        tower.trunk.zero_grad()
        res.append((float(grad_cache_loss(tower, q, tower, d, 4, scale)), tower.trunk.flat_grad.clone()))
    assert res[0][0] == res[1][0]
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-4 * float(res[0][1].abs().max())
    tower = _tower(0.0, 0.0, attn=0.15).train()
    with torch.no_grad():
        e_train = tower(**q)["embedding"].clone()
        tower.eval()
        e_eval = tower(**q)["embedding"].clone()
        tower.train()
    assert not torch.equal(e_train, e_eval) and rel_err(e_train, e_eval) < 0.5
    torch.manual_seed(78)
    tower.trunk.zero_grad()
    l0 = float(grad_cache_loss(tower, q, tower, d, 4, scale))
    gvec = tower.trunk.flat_grad.clone()
    lo, hi = tower.trunk._layout["encoder.layers.0.attn.Wqkv.weight"][0], tower.trunk.n_decay
    direction = torch.zeros_like(gvec)
    direction[lo:hi] = gvec[lo:hi]
    eps = 2e-2 / float(direction.norm())
    with torch.no_grad():
        tower.trunk.flat_param.add_(direction, alpha=eps)
    tower.trunk.sync_shadows()
    torch.manual_seed(78)
    tower.trunk.zero_grad()
    l1 = float(grad_cache_loss(tower, q, tower, d, 4, scale))
    predicted = eps * float((direction * gvec).sum())
    report("attn_dropout_gradcache", l0=l0, l1=l1, predicted=predicted, actual=l1 - l0)
    assert predicted > 0 and abs((l1 - l0) - predicted) < 0.35 * predicted + 2e-3


def test_single_pass_dropout_attention_equals_the_general_kernels():
    """Round 4: max_seqlen <= 128 with attn_pdrop > 0 runs the <DROP> instantiations of the single-pass forward and of the
    fused persistent backward (the reference's bert-base-uncased recipes train with p = 0.1).  Same Philox mask, same
    arithmetic as the general streaming kernels (cx_attn_set_fwd_s128(0) / cx_attn_set_bwd_s128(0) in the dev library):
    outputs to bf16 rounding of a differently ordered sum, gradients likewise."""
    lib = _C.dev_lib()
    H, D, p = 12, 64, 0.1
    lens = [128, 128, 96, 17, 128]
    B, T = len(lens), sum(lens)
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(T, 3 * H * D, generator=g) * 0.7).to(DEV).bfloat16()
    dout = torch.randn(T, H * D, generator=g).to(DEV).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    inv = 1.0 / (1000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.outer(torch.arange(128, dtype=torch.float32), inv)
    cos, sin = torch.cos(fr).to(DEV).contiguous(), torch.sin(fr).to(DEV).contiguous()
    res = {}
    try:
        for mode in (2, 0):
            lib.cx_attn_set_fwd_s128(mode)
            lib.cx_attn_set_bwd_s128(3 if mode else 0)
            out = torch.empty(T, H * D, device=DEV, dtype=torch.bfloat16)
            lse = torch.empty(H * T, device=DEV)
            dqkv = torch.zeros_like(qkv)
            delta = torch.empty(H * T, device=DEV)
            _C.check(lib.cx_attn_varlen_dropout_fwd(qkv.data_ptr(), cu.data_ptr(), cos.data_ptr(), sin.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                                    B, H, T, 128, 0.125, p, 99, 7, 3, _C.cur_stream()))
            _C.check(lib.cx_attn_varlen_dropout_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), cu.data_ptr(), cos.data_ptr(),
                                                    sin.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, H, T, 128, 0.125, p, 99, 7, 3,
                                                    _C.cur_stream()))
            torch.cuda.synchronize()
            res[mode] = (out, lse, dqkv)
    finally:
        lib.cx_attn_set_fwd_s128(2)
        lib.cx_attn_set_bwd_s128(3)
    e_o = rel_err(res[2][0].float(), res[0][0].float())
    e_l = rel_err(res[2][1], res[0][1])
    e_g = rel_err(res[2][2].float(), res[0][2].float())
    report("attn_dropout_single_pass_vs_general", e_out=e_o, e_lse=e_l, e_dqkv=e_g)
    assert e_o < 4e-3 and e_l < 1e-5 and e_g < 8e-3
    # the backward of the single-pass pair is deterministic
    assert torch.isfinite(res[2][2].float()).all()
