"""Level-0 drop-in on silicon: the reference's flash path re-declared from the ops it imports from
`flash_attn` -- unpad_input, FusedDense, apply_rotary_emb_func (varlen), flash_attn_varlen_qkvpacked_func,
dropout_add_layer_norm (fp32 residual for layer 0), swiglu, pad_input -- each bound to the HIP backend through
contrastors_amd.flash_attn_api.  The composition follows sc/models/encoder/modeling_nomic_bert.py:515-587,307-395,
sc/layers/block.py:389-463 (post-norm), sc/layers/attention.py:112-135,172-182,243 and sc/layers/mlp.py:68-83 line by line
(re-declared here because /root/reference does not exist on the GPU box; tests/test_shim_cpu.py imports and constructs the
reference's own modules on this surface in the build container).  Judged against the fp32 oracle with the reference's
3 x bf16-eager rule, and against the native engine (the same arithmetic fused into one call)."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

from contrastors_amd.flash_attn_api.bert_padding import pad_input, unpad_input
from contrastors_amd.flash_attn_api.flash_attn_interface import flash_attn_varlen_qkvpacked_func
from contrastors_amd.flash_attn_api.layers.rotary import apply_rotary_emb_func
from contrastors_amd.flash_attn_api.ops.activations import swiglu
from contrastors_amd.flash_attn_api.ops.fused_dense import fused_dense_func
from contrastors_amd.flash_attn_api.ops.layer_norm import dropout_add_layer_norm, layer_norm
from contrastors_amd.nomic_bert import NomicBertConfig, NomicBertEngine, VarlenBatch
from oracle import encoder_ref
from tests.gpu_util import max_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _shim_forward(sd, cfg, ids, mask):
    """NomicBertModel.forward on the shim ops, bf16 autocast semantics of the reference (fp32 embeddings + embedding LN,
    fp32 residual into layer 0, bf16 afterwards)."""
    d, H = cfg.n_embd, cfg.n_head
    emb = sd["embeddings.word_embeddings.weight"][ids] + sd["embeddings.token_type_embeddings.weight"][0]
    hidden = layer_norm(emb, sd["emb_ln.weight"], sd["emb_ln.bias"], cfg.layer_norm_epsilon)      # fp32 in -> fp32 out
    B, S = ids.shape
    hidden, indices, cu, max_s = unpad_input(hidden, mask)[:4]                                     # (T, d)
    inv_freq = 1.0 / (cfg.rotary_emb_base ** (torch.arange(0, 64, 2, device=DEV, dtype=torch.float32) / 64))
    freqs = torch.outer(torch.arange(S, device=DEV, dtype=torch.float32), inv_freq)
    cos, sin = freqs.cos().to(torch.bfloat16), freqs.sin().to(torch.bfloat16)                      # quirk 12: cast tables
    for l in range(cfg.n_layer):
        p = f"encoder.layers.{l}."
        x16 = hidden.to(torch.bfloat16)
        qkv = fused_dense_func(x16, sd[p + "attn.Wqkv.weight"]).view(-1, 3, H, 64)
        q = apply_rotary_emb_func(qkv[:, 0].contiguous(), cos, sin, cu_seqlens=cu, max_seqlen=max_s)
        k = apply_rotary_emb_func(qkv[:, 1].contiguous(), cos, sin, cu_seqlens=cu, max_seqlen=max_s)
        qkv = torch.stack([q, k, qkv[:, 2]], dim=1)                                                # embedding.py:706
        ctx = flash_attn_varlen_qkvpacked_func(qkv, cu, max_s, 0.0, softmax_scale=torch.tensor(1.0 / 8.0), causal=False)
        attn_out = fused_dense_func(ctx.reshape(-1, d), sd[p + "attn.out_proj.weight"])
        # block.py:422-431: fused dropout(0) + add + LayerNorm; layer 0 carries the fp32 residual
        hidden = dropout_add_layer_norm(attn_out, hidden, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 0.0,
                                        cfg.layer_norm_epsilon, prenorm=False, residual_in_fp32=(l == 0))
        y = fused_dense_func(hidden, sd[p + "mlp.fc11.weight"])
        gate = fused_dense_func(hidden, sd[p + "mlp.fc12.weight"])
        mlp_out = fused_dense_func(swiglu(gate, y), sd[p + "mlp.fc2.weight"])                      # mlp.py:75
        hidden = dropout_add_layer_norm(mlp_out, hidden, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 0.0,
                                        cfg.layer_norm_epsilon, prenorm=False)
    return pad_input(hidden, indices, B, S)                                                        # (B, S, d), pads zero


def test_reference_flash_path_composed_from_shim_ops_matches_oracle_and_engine():
    cfg = NomicBertConfig(vocab_size=1024, n_embd=256, n_head=4, n_inner=512, n_layer=2, n_positions=256)
    ns = SimpleNamespace(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    sd = {k: v.to(DEV) for k, v in encoder_ref.random_state_dict(ns, 4).items()}
    g = torch.Generator().manual_seed(5)
    B, S = 6, 96
    lens = torch.randint(20, S + 1, (B,), generator=g)
    lens[0] = S
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    ids = (torch.randint(3, 1024, (B, S), generator=g) * mask).to(DEV)
    mask = mask.to(DEV)
    hid = _shim_forward(sd, cfg, ids, mask).float()
    ref = encoder_ref.encoder_hidden_states(sd, ns, ids, mask) * mask[..., None]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref16 = encoder_ref.encoder_hidden_states(sd, ns, ids, mask).float() * mask[..., None]
    e_shim, e_b16 = max_err(hid, ref), max_err(ref16, ref)
    assert torch.count_nonzero(hid * (1 - mask[..., None])) == 0          # pad_input zero-fills (modeling_nomic_bert.py:392)
    # the native engine fuses the very same arithmetic: mean-pooled embeddings must agree closely
    eng = NomicBertEngine(cfg, device=DEV)
    eng.load_reference_state_dict(sd)
    emb_eng, _ = eng.forward_chunk(VarlenBatch.from_lengths(ids, lens.numpy()), False)
    pooled = (hid * mask[..., None]).sum(1) / mask.sum(1, keepdim=True)
    e_eng = max_err(F.normalize(pooled, dim=-1), emb_eng)
    report("shim_compose", e_shim=e_shim, e_bf16_eager=e_b16, e_vs_engine=e_eng)
    assert e_shim <= 3 * e_b16 + 1e-3, (e_shim, e_b16)
    assert e_eng < 5e-3


def test_layer_norm_shim_keeps_fp32_residual_and_inputs():
    """`residual_in_fp32` / fp32 operands: out in x0's dtype, z in fp32, gradients in the inputs' dtypes."""
    d = 768
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(37, d, generator=g).to(DEV).to(torch.bfloat16).requires_grad_()
    res = torch.randn(37, d, generator=g).to(DEV).requires_grad_()                   # fp32 residual (layer 0)
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV).requires_grad_()
    b = (0.1 * torch.randn(d, generator=g)).to(DEV).requires_grad_()
    out, z = dropout_add_layer_norm(x0, res, w, b, 0.0, 1e-12, prenorm=True, residual_in_fp32=True)
    assert out.dtype == torch.bfloat16 and z.dtype == torch.float32
    assert torch.equal(z, x0.detach().float() + res.detach())                          # the sum is exact in fp32
    xr, rr = x0.detach().float().requires_grad_(), res.detach().clone().requires_grad_()
    wr, br = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    ref = F.layer_norm(xr + rr, (d,), wr, br, 1e-12)
    go, gz = torch.randn(37, d, generator=g).to(DEV), torch.randn(37, d, generator=g).to(DEV)
    torch.autograd.backward([out, z], [go.to(torch.bfloat16), gz])
    torch.autograd.backward([ref, xr + rr], [go.to(torch.bfloat16).float(), gz])
    assert res.grad.dtype == torch.float32 and x0.grad.dtype == torch.bfloat16
    assert max_err(out.float(), ref) < 0.04 and max_err(res.grad, rr.grad) < 2e-4 * float(rr.grad.abs().max()) + 1e-5
    assert max_err(w.grad, wr.grad) < 1e-3 * float(wr.grad.abs().max())
    # fp32 in -> fp32 out (the embedding LayerNorm of the BERT path)
    x32 = torch.randn(11, d, generator=g).to(DEV)
    y = layer_norm(x32, w.detach(), b.detach(), 1e-12)
    assert y.dtype == torch.float32 and max_err(y, F.layer_norm(x32, (d,), w.detach(), b.detach(), 1e-12)) < 1e-5
