"""The flash_attn-surface ops (contrastors_amd.flash_attn_api) through torch autograd vs plain torch references."""
import math

import numpy as np
import pytest
import torch

import contrastors_amd.flash_attn_api as fa
from contrastors_amd.flash_attn_api.layers.rotary import RotaryEmbedding, apply_rotary_emb_func, apply_rotary_emb_qkv_
from contrastors_amd.flash_attn_api.ops.activations import swiglu
from contrastors_amd.flash_attn_api.ops.fused_dense import FusedDense
from contrastors_amd.flash_attn_api.ops.layer_norm import dropout_add_layer_norm, layer_norm
from oracle import encoder_ref
from tests.gpu_util import max_err, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _r(*s, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * std).to(DEV)


def test_varlen_attention_autograd():
    lens, H = [100, 128, 37], 4
    T = sum(lens)
    qkv = _r(T, 3, H, 64, seed=1).to(torch.bfloat16).requires_grad_()
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    scale = torch.tensor(1.0 / math.sqrt(64.0), device=DEV)  # contrastors passes a 0-dim tensor
    out = fa.flash_attn_varlen_qkvpacked_func(qkv, cu, max(lens), 0.0, softmax_scale=scale, causal=False)
    do = _r(T, H, 64, seed=2).to(torch.bfloat16)
    out.backward(do)
    ref_in = qkv.detach().float().requires_grad_()
    outs, t0 = [], 0
    for l in lens:
        x = ref_in[t0:t0 + l]
        s = torch.einsum("qhd,khd->hqk", x[:, 0], x[:, 1]) / 8.0
        outs.append(torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), x[:, 2]))
        t0 += l
    ref = torch.cat(outs)
    ref.backward(do.float())
    assert rel_err(out.float(), ref) < 6e-3 and rel_err(qkv.grad.float(), ref_in.grad) < 1.5e-2


def test_fixed_length_attention_vit_shape():
    qkv = _r(3, 197, 3, 2, 64, seed=3).to(torch.bfloat16)
    out = fa.flash_attn_qkvpacked_func(qkv)
    q, k, v = qkv.float().unbind(2)
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q, k) / 8.0, -1), v)
    assert out.shape == (3, 197, 2, 64) and rel_err(out.float(), ref) < 6e-3


@pytest.mark.parametrize("k_in,n_out,bias", [(768, 2304, False), (100, 36, True)])
def test_fused_dense_matches_linear(k_in, n_out, bias):
    torch.manual_seed(0)
    fd = FusedDense(k_in, n_out, bias=bias).to(DEV)
    ref = torch.nn.Linear(k_in, n_out, bias=bias).to(DEV)
    ref.load_state_dict(fd.state_dict())
    x = _r(5, 40, k_in, seed=4).to(torch.bfloat16).requires_grad_()
    xr = x.detach().float().requires_grad_()
    y = fd(x)
    yr = ref(xr)
    g = _r(5, 40, n_out, seed=5)
    y.backward(g.to(torch.bfloat16))
    yr.backward(g.to(torch.bfloat16).float())
    assert y.dtype == torch.bfloat16 and rel_err(y.float(), yr) < 5e-3
    assert rel_err(x.grad.float(), xr.grad) < 8e-3
    assert rel_err(fd.weight.grad, ref.weight.grad) < 8e-3
    if bias:
        assert rel_err(fd.bias.grad, ref.bias.grad) < 5e-3


@pytest.mark.parametrize("prenorm", [False, True])
def test_dropout_add_layer_norm(prenorm):
    d = 768
    x0 = _r(4, 33, d, seed=6).to(torch.bfloat16).requires_grad_()
    res = _r(4, 33, d, seed=7).to(torch.bfloat16).requires_grad_()
    w = (1 + _r(d, seed=8, std=0.1)).requires_grad_()
    b = _r(d, seed=9, std=0.1).requires_grad_()
    out = dropout_add_layer_norm(x0, res, w, b, 0.0, 1e-12, prenorm=prenorm)
    xr, rr = x0.detach().float().requires_grad_(), res.detach().float().requires_grad_()
    wr, br = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    z = xr + rr
    ref = torch.nn.functional.layer_norm(z, (d,), wr, br, 1e-12)
    g = _r(4, 33, d, seed=10).to(torch.bfloat16)
    if prenorm:
        o, zz = out
        g2 = _r(4, 33, d, seed=11).to(torch.bfloat16)
        (o.float() * g.float()).sum().backward(retain_graph=True)
        # second use of the residual stream: gradient flowing into z
        x0.grad = res.grad = None
        w.grad = b.grad = None
        torch.autograd.backward([o, zz], [g, g2])
        torch.autograd.backward([ref, z], [g.float(), g2.float()])
        assert rel_err(zz.float(), z) < 4e-3
    else:
        o = out
        o.backward(g)
        ref.backward(g.float())
    assert rel_err(o.float(), ref) < 4e-3
    assert rel_err(x0.grad.float(), xr.grad) < 1e-2 and rel_err(res.grad.float(), rr.grad) < 1e-2
    assert rel_err(w.grad, wr.grad) < 1e-2 and rel_err(b.grad, br.grad) < 1e-2
    y = layer_norm(x0.detach(), w.detach(), b.detach(), 1e-12)
    assert rel_err(y.float(), torch.nn.functional.layer_norm(x0.detach().float(), (d,), w.detach(), b.detach(), 1e-12)) < 4e-3


@pytest.mark.parametrize("x0_dtype,res_fp32", [(torch.bfloat16, False), (torch.float32, True)])
def test_dropout_add_layer_norm_with_dropout(x0_dtype, res_fp32):
    """p > 0 (flash_attn.ops.layer_norm.dropout_add_layer_norm, sc/layers/block.py:422-431 with resid_pdrop > 0): the
    mask comes from the device generator's Philox stream -- reproducible under torch.manual_seed, replayed by RandContext
    -- and out / gradients equal torch's LayerNorm of x0 * mask / (1 - p) + residual with THAT mask."""
    from contrastors_amd.rand_state import RandContext

    d, p = 768, 0.1
    x0 = _r(4, 33, d, seed=6).to(x0_dtype).requires_grad_()
    res = _r(4, 33, d, seed=7).to(torch.float32 if res_fp32 else torch.bfloat16).requires_grad_()
    w = (1 + _r(d, seed=8, std=0.1)).requires_grad_()
    b = _r(d, seed=9, std=0.1).requires_grad_()
    torch.manual_seed(5)
    ctx = RandContext([x0])
    o, mask = dropout_add_layer_norm(x0, res, w, b, p, 1e-12, residual_in_fp32=res_fp32, return_dropout_mask=True)
    rate = mask.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01, rate
    g = _r(4, 33, d, seed=10).to(o.dtype)
    o.backward(g)
    xr, rr = x0.detach().float().requires_grad_(), res.detach().float().requires_grad_()
    wr, br = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    ref = torch.nn.functional.layer_norm(xr * mask.float() / (1 - p) + rr, (d,), wr, br, 1e-12)
    ref.backward(g.float())
    assert rel_err(o.float(), ref) < 4e-3
    assert rel_err(x0.grad.float(), xr.grad) < 1e-2 and rel_err(res.grad.float(), rr.grad) < 1e-2
    assert rel_err(w.grad, wr.grad) < 1e-2 and rel_err(b.grad, br.grad) < 1e-2
    assert (x0.grad[~mask] == 0).all()
    # same generator state -> same mask; a later call draws a different one; RandContext replays the first
    torch.manual_seed(5)
    _, m2 = dropout_add_layer_norm(x0.detach(), res.detach(), w.detach(), b.detach(), p, 1e-12, residual_in_fp32=res_fp32,
                                   return_dropout_mask=True)
    _, m3 = dropout_add_layer_norm(x0.detach(), res.detach(), w.detach(), b.detach(), p, 1e-12, residual_in_fp32=res_fp32,
                                   return_dropout_mask=True)
    assert torch.equal(m2, mask) and not torch.equal(m3, mask)
    with ctx:
        _, m4 = dropout_add_layer_norm(x0.detach(), res.detach(), w.detach(), b.detach(), p, 1e-12,
                                       residual_in_fp32=res_fp32, return_dropout_mask=True)
    assert torch.equal(m4, mask)
    # eval-style call (p = 0) is the plain op
    o0 = dropout_add_layer_norm(x0.detach(), res.detach(), w.detach(), b.detach(), 0.0, 1e-12, residual_in_fp32=res_fp32)
    assert rel_err(o0.float(), torch.nn.functional.layer_norm(x0.detach().float() + res.detach().float(), (d,), w.detach(),
                                                              b.detach(), 1e-12)) < 4e-3


def test_swiglu_and_rotary():
    x = _r(7, 50, 512, seed=12).to(torch.bfloat16).requires_grad_()
    y = _r(7, 50, 512, seed=13).to(torch.bfloat16).requires_grad_()
    out = swiglu(x, y)
    xr, yr = x.detach().float().requires_grad_(), y.detach().float().requires_grad_()
    ref = torch.nn.functional.silu(xr) * yr
    g = _r(7, 50, 512, seed=14).to(torch.bfloat16)
    out.backward(g)
    ref.backward(g.float())
    assert rel_err(out.float(), ref) < 4e-3 and rel_err(x.grad.float(), xr.grad) < 6e-3
    assert rel_err(y.grad.float(), yr.grad) < 6e-3
    # varlen rotary on q (T,H,64): positions restart per sequence
    lens, H = [40, 128, 9], 3
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    cos, sin = encoder_ref.rotary_tables(128, 64, 1000.0)
    cos, sin = cos.to(DEV), sin.to(DEV)
    q = _r(T, H, 64, seed=15).to(torch.bfloat16).requires_grad_()
    qo = apply_rotary_emb_func(q, cos.to(torch.bfloat16), sin.to(torch.bfloat16), False, False, 0, cu, 128)
    parts, t0 = [], 0
    c16, s16 = cos.to(torch.bfloat16).float(), sin.to(torch.bfloat16).float()  # reference casts tables to bf16 (quirk 12)
    qr = q.detach().float().requires_grad_()
    for l in lens:
        parts.append(encoder_ref.apply_rotary(qr[t0:t0 + l].unsqueeze(0), c16, s16)[0])
        t0 += l
    ref = torch.cat(parts)
    gq = _r(T, H, 64, seed=16).to(torch.bfloat16)
    qo.backward(gq)
    ref.backward(gq.float())
    assert rel_err(qo.float(), ref) < 4e-3 and rel_err(q.grad.float(), qr.grad) < 6e-3
    # packed (B,S,3,H,64) in-place form + the module
    qkv = _r(2, 64, 3, H, 64, seed=17).to(torch.bfloat16)
    keep = qkv.clone()
    rot = RotaryEmbedding(64, base=1000.0).to(DEV)
    out = rot(qkv)
    assert out.data_ptr() == qkv.data_ptr() and torch.equal(qkv[:, :, 2], keep[:, :, 2])
    want_q = encoder_ref.apply_rotary(keep[:, :, 0].float(), c16, s16)
    assert rel_err(qkv[:, :, 0].float(), want_q) < 4e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N,V,inplace", [(1000, 30528, True), (37, 30522, False), (64, 512, False)])
def test_cross_entropy_k12(dtype, N, V, inplace):
    """flash_attn.losses.cross_entropy.CrossEntropyLoss on cx_xent_fwd/bwd vs torch fp32 (ignore_index rows included)."""
    from contrastors_amd.flash_attn_api.losses.cross_entropy import CrossEntropyLoss

    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(N, V, generator=g) * 3).to(DEV).to(dtype)
    labels = torch.randint(0, V, (N,), generator=g)
    labels[::7] = -100
    labels = labels.to(DEV)
    ref_in = logits.float().clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(ref_in, labels, ignore_index=-100)
    ref.backward()
    x = logits.clone().requires_grad_()
    xin = x * 1.0  # a non-leaf the in-place backward is allowed to overwrite
    loss = CrossEntropyLoss(inplace_backward=inplace)(xin, labels)
    loss.backward()
    assert abs(float(loss) - float(ref)) < (2e-3 if dtype == torch.bfloat16 else 1e-5) * max(1.0, abs(float(ref)))
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(x.grad.float(), ref_in.grad) < tol
    assert float(x.grad[::7].abs().max()) == 0.0
    per_row = CrossEntropyLoss(reduction="none")(logits, labels)
    want = torch.nn.functional.cross_entropy(logits.float(), labels, ignore_index=-100, reduction="none")
    assert max_err(per_row, want) < (5e-2 if dtype == torch.bfloat16 else 1e-4)
    with pytest.raises(NotImplementedError):
        CrossEntropyLoss(label_smoothing=0.1)


@pytest.mark.parametrize("prenorm,with_res", [(False, False), (False, True), (True, True)])
def test_rms_norm_k8(prenorm, with_res):
    """flash_attn.ops.rms_norm (K8): rms_norm / dropout_add_rms_norm / RMSNorm vs torch in fp32."""
    from contrastors_amd.flash_attn_api.ops.rms_norm import RMSNorm, dropout_add_rms_norm, rms_norm

    d, eps = 768, 1e-6
    x0 = _r(3, 41, d, seed=21).to(torch.bfloat16).requires_grad_()
    res = _r(3, 41, d, seed=22).to(torch.bfloat16).requires_grad_() if with_res else None
    w = (1 + _r(d, seed=23, std=0.1)).requires_grad_()
    out = dropout_add_rms_norm(x0, res, w, None, 0.0, eps, prenorm=prenorm)
    xr = x0.detach().float().requires_grad_()
    rr = res.detach().float().requires_grad_() if with_res else None
    wr = w.detach().clone().requires_grad_()
    z = xr + rr if with_res else xr
    ref = z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + eps) * wr
    g = _r(3, 41, d, seed=24).to(torch.bfloat16)
    if prenorm:
        o, zz = out
        g2 = _r(3, 41, d, seed=25).to(torch.bfloat16)
        torch.autograd.backward([o, zz], [g, g2])
        torch.autograd.backward([ref, z], [g.float(), g2.float()])
        assert rel_err(zz.float(), z) < 4e-3
    else:
        o = out
        o.backward(g)
        ref.backward(g.float())
    assert rel_err(o.float(), ref) < 4e-3
    assert rel_err(x0.grad.float(), xr.grad) < 1e-2 and rel_err(w.grad, wr.grad) < 1e-2
    if with_res:
        assert rel_err(res.grad.float(), rr.grad) < 1e-2
    m = RMSNorm(d, eps=eps).to(DEV)
    with torch.no_grad():
        m.weight.copy_(w.detach())
    y = m(x0.detach())
    assert m.bias is None and rel_err(y.float(), (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps) * wr).detach()) < 4e-3
    assert torch.equal(y, rms_norm(x0.detach(), w.detach(), eps))
