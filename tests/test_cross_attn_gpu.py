"""K3: kv-packed cross-attention (flash_attn_kvpacked_func / flash_attn_varlen_kvpacked_func; the reference's
FlashAttentionPooling, sc/layers/attention.py:313-433) through torch autograd vs plain fp32 torch attention, and the
pooling module's own composition (latent query -> Wq, Wkv -> cross-attention -> out_proj) re-declared on the shim ops."""
import math

import pytest
import torch

import contrastors_amd.flash_attn_api as fa
from contrastors_amd.flash_attn_api.bert_padding import pad_input, unpad_input
from tests.gpu_util import rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref_cross(q, k, v, scale):
    """q (Sq,H,D), k/v (Sk,H,D) fp32 -> (Sq,H,D)"""
    s = torch.einsum("qhd,khd->hqk", q, k) * scale
    return torch.einsum("hqk,khd->qhd", torch.softmax(s, dim=-1), v)


@pytest.mark.parametrize("lens_q,lens_k", [
    ([1, 1, 1, 1], [128, 37, 1, 200]),          # the pooling use: one latent query per sequence, ragged keys
    ([5, 130, 64, 257], [300, 64, 129, 31]),    # general rectangular problems, several query / key tiles
    ([3, 0, 7], [10, 20, 0]),                   # empty sides: no queries (dk = dv = 0), no keys (out = 0, dq = 0)
])
def test_varlen_kvpacked_matches_torch(lens_q, lens_k):
    H, D = 12, 64
    g = torch.Generator().manual_seed(3)
    Tq, Tk = sum(lens_q), sum(lens_k)
    q = (torch.randn(max(Tq, 1), H, D, generator=g) * 0.8)[:Tq].to(DEV).bfloat16().requires_grad_()
    kv = (torch.randn(max(Tk, 1), 2, H, D, generator=g) * 0.8)[:Tk].to(DEV).bfloat16().requires_grad_()
    cu_q = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32, device=DEV)
    cu_k = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(D)
    out = fa.flash_attn_varlen_kvpacked_func(q, kv, cu_q, cu_k, max(lens_q), max(lens_k), 0.0, softmax_scale=scale)
    go = (torch.randn(max(Tq, 1), H, D, generator=g)[:Tq]).to(DEV).bfloat16()
    out.backward(go)

    qr, kvr = q.detach().float().requires_grad_(), kv.detach().float().requires_grad_()
    outs = []
    for b in range(len(lens_q)):
        qs = qr[cu_q[b]:cu_q[b + 1]]
        ks = kvr[cu_k[b]:cu_k[b + 1]]
        if qs.shape[0] == 0:
            continue
        outs.append(_ref_cross(qs, ks[:, 0], ks[:, 1], scale) if ks.shape[0] else torch.zeros_like(qs))
    ref = torch.cat(outs) if outs else qr * 0
    ref.backward(go.float())
    assert torch.isfinite(out).all() and torch.isfinite(q.grad).all() and torch.isfinite(kv.grad).all()
    e_o, e_q, e_kv = rel_err(out.float(), ref), rel_err(q.grad.float(), qr.grad), rel_err(kv.grad.float(), kvr.grad)
    report("cross_attn", lens_q=str(lens_q), e_out=e_o, e_dq=e_q, e_dkv=e_kv)
    assert e_o < 1e-2 and e_q < 2e-2 and e_kv < 2e-2


def test_fixed_length_kvpacked_and_self_attention_consistency():
    """(B,Sq,H,D) x (B,Sk,2,H,D); with q = the sequence's own queries it must equal the qkv-packed self-attention op."""
    B, S, H, D = 3, 197, 12, 64
    g = torch.Generator().manual_seed(4)
    qkv = (torch.randn(B, S, 3, H, D, generator=g) * 0.7).to(DEV).bfloat16()
    o_self = fa.flash_attn_qkvpacked_func(qkv, 0.0)
    o_cross = fa.flash_attn_kvpacked_func(qkv[:, :, 0].contiguous(), qkv[:, :, 1:].contiguous(), 0.0)
    assert o_cross.shape == (B, S, H, D)
    assert rel_err(o_cross.float(), o_self.float()) < 4e-3


class _Pooling(torch.nn.Module):
    """FlashAttentionPooling re-declared on the shim's ops (attention.py:325-433): latent -> Wq; kv -> Wkv; unpad;
    flash_attn_varlen_kvpacked_func; pad; out_proj."""

    def __init__(self, d, H):
        super().__init__()
        from contrastors_amd.flash_attn_api.ops.fused_dense import FusedDense

        self.H, self.hd = H, d // H
        self.Wq, self.Wkv, self.out_proj = FusedDense(d, d), FusedDense(d, 2 * d), FusedDense(d, d)
        self.latent = torch.nn.Parameter(torch.randn(1, 1, d) * d ** -0.5)

    def forward(self, x, attention_mask):
        B = x.shape[0]
        q = self.Wq(self.latent.expand(B, -1, -1).to(x.dtype)).view(B, 1, self.H, self.hd)
        kv = self.Wkv(x).view(B, x.shape[1], 2, self.H, self.hd)
        uq, _, cu_q, max_q = unpad_input(q, torch.ones(B, 1, dtype=attention_mask.dtype, device=x.device))[:4]
        ukv, _, cu_k, max_k = unpad_input(kv, attention_mask)[:4]
        o = fa.flash_attn_varlen_kvpacked_func(uq, ukv, cu_q, cu_k, max_q, max_k, 0.0, softmax_scale=1.0 / math.sqrt(self.hd))
        return self.out_proj(o.reshape(B, 1, -1))


def test_attention_pooling_module_composition():
    d, H, B, S = 768, 12, 6, 50
    torch.manual_seed(0)
    pool = _Pooling(d, H).to(DEV).to(torch.bfloat16)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, S, d, generator=g).to(DEV).bfloat16().requires_grad_()
    lens = torch.tensor([50, 1, 17, 33, 50, 8])
    mask = (torch.arange(S)[None] < lens[:, None]).to(DEV).long()
    out = pool(x, mask)
    go = torch.randn(B, 1, d, generator=g).to(DEV).bfloat16()
    out.backward(go)

    xr = x.detach().float().requires_grad_()
    W = {n: p.detach().float().requires_grad_() for n, p in pool.named_parameters()}
    q = (W["latent"].expand(B, -1, -1) @ W["Wq.weight"].T + W["Wq.bias"]).view(B, 1, H, d // H)
    kv = (xr @ W["Wkv.weight"].T + W["Wkv.bias"]).view(B, S, 2, H, d // H)
    outs = []
    for b in range(B):
        n = int(lens[b])
        outs.append(_ref_cross(q[b], kv[b, :n, 0], kv[b, :n, 1], 1.0 / math.sqrt(d // H)).reshape(1, d))
    ref = torch.stack(outs) @ W["out_proj.weight"].T + W["out_proj.bias"]
    ref.backward(go.float())
    assert rel_err(out.float(), ref) < 2e-2
    assert rel_err(x.grad.float(), xr.grad) < 3e-2
    for n, p in pool.named_parameters():
        assert rel_err(p.grad.float(), W[n].grad) < 3e-2, n
    assert (x.grad[1, 1:] == 0).all()      # padded keys get no gradient
