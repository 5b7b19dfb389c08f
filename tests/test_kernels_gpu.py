"""Per-kernel parity: every C-ABI entry point vs a plain torch fp32 reference of the same op, on seeded inputs.
Tolerances are written at each assert.  Runs on the MI355X box only (-m gpu)."""
import math

import numpy as np
import pytest
import torch

from contrastors_amd import _C
from tests.gpu_util import LD, L, S, bf, gemm, max_err, rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _randn(*s, seed=0, std=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*s, generator=g) * std).to(DEV)


# ------------------------------------------------------------------------------------------------- hardware probes
def test_probe_mfma_accumulator_layout():
    out = torch.zeros(32, 32, device=DEV)
    _C.check(LD().cx_probe_mfma_layout(out.data_ptr(), S()))
    i = torch.arange(32, device=DEV).float()
    want = (i[:, None] + 1) + 64 * (i[None, :] + 1)  # asymmetric: catches a transposed C mapping
    assert torch.equal(out, want)


def test_probe_ds_read_tr16_pattern():
    src = torch.arange(256, dtype=torch.int16, device=DEV)
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    _C.check(LD().cx_probe_ds_read_tr16(src.data_ptr(), out.data_ptr(), S()))
    got = out.cpu().numpy().reshape(64, 4)
    want = np.zeros((64, 4), dtype=np.int16)
    for lane in range(64):
        g, i = lane // 16, lane % 16
        for j in range(4):
            want[lane, j] = (g * 16 + 4 * j + i // 4) * 4 + (i % 4)
    report("probe_tr16", match=bool((got == want).all()), got_lane0=got[0].tolist(), got_lane1=got[1].tolist(),
           got_lane5=got[5].tolist(), got_lane17=got[17].tolist())
    if not (got == want).all():
        pytest.xfail("ds_read_b64_tr_b16 lane pattern differs from the documented one (not used by shipped kernels)")


# ------------------------------------------------------------------------------------------------------------ GEMM
def _gemm_lib(variant):
    """"product" -> libcontrastors_hip.so (one kernel family, no switches); anything else -> the dev library with its
    process-global variant switch (earlier GEMM generations kept for A/B)."""
    if variant in ("product", 6):
        return L()
    lib = LD()
    lib.cx_gemm_set_variant({"v6": 6, "v5": 5, "v2": 2}.get(variant, variant if isinstance(variant, int) else 1))
    lib.cx_gemm_set_glds(0 if variant == "v1_reg" else 1)
    return lib


def _gemm_lib_reset():
    LD().cx_gemm_set_variant(6)
    LD().cx_gemm_set_glds(1)


@pytest.mark.parametrize("variant", ["product", "v6", "v5", "v2", "v1_glds", "v1_reg"])
@pytest.mark.parametrize("M,N,K", [(8192, 2304, 768), (8192, 768, 3072), (1000, 768, 768), (130, 132, 64),
                                   (257, 6144, 768), (300, 768, 128)])
def test_gemm_bf16_nt(variant, M, N, K):
    lib = _gemm_lib(variant)
    try:
        x, w = bf(_randn(M, K, seed=1)), bf(_randn(N, K, seed=2, std=0.05))
        bias = _randn(N, seed=3)
        ref = x.float() @ w.float().T
        out32 = gemm(x, w, out_mode=1, lib=lib)
        e32 = rel_err(out32, ref)
        out16 = gemm(x, w, bias=bias, out_mode=0, lib=lib)
        e16 = rel_err(out16.float(), ref + bias)
        acc = torch.ones(M, N, device=DEV)
        if variant == "product":  # no atomically-accumulating mode in the product library: the caller is told so
            assert lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), acc.data_ptr(), None, M, N, K, K, K, N, 2, 4, 1.0, S()) == -3
            eacc = 0.0
        else:
            gemm(x, w, out_mode=2, split_k=min(4, K // 64), out=acc, lib=lib)
            eacc = rel_err(acc - 1.0, ref)
        # workspace split-K accumulate (the wgrad form): Out += X W^T, twice, deterministic
        acc2 = torch.ones(M, N, device=DEV)
        ws = torch.empty(3 * M * N + 5, device=DEV)
        for _ in range(2):
            _C.check(lib.cx_gemm_bf16_nt_accum(x.data_ptr(), w.data_ptr(), acc2.data_ptr(), ws.data_ptr(), ws.numel(),
                                               M, N, K, K, K, S()))
        eacc2 = rel_err(acc2 - 1.0, 2 * ref)
        report("gemm", variant=variant, M=M, N=N, K=K, e32=e32, e16=e16, eacc=eacc, eacc2=eacc2)
        assert e32 < 1e-5, "fp32-out GEMM: only accumulation-order error allowed"
        assert e16 < 4e-3, "bf16-out GEMM: one bf16 rounding (2^-9 rel) of the fp32 result"
        assert eacc < 1e-5 and eacc2 < 1e-5
    finally:
        _gemm_lib_reset()


@pytest.mark.parametrize("variant", [5, 6])
@pytest.mark.parametrize("M,I,K", [(8192, 3072, 768), (300, 512, 256), (257, 96, 64)])
def test_gemm_swiglu_fused(M, I, K, variant):
    """fc11 || fc12 GEMM with SwiGLU in the epilogue == standalone GEMM + swiglu (interleaved-by-32 weight rows)."""
    lib = _gemm_lib(variant)  # 6 -> product library; 5 -> the v5 kernel's fused epilogue (dev library)
    x = bf(_randn(M, K, seed=90))
    w11, w12 = bf(_randn(I, K, seed=91, std=0.05)), bf(_randn(I, K, seed=92, std=0.05))
    wi = torch.stack([w11.view(I // 32, 32, K), w12.view(I // 32, 32, K)], 1).reshape(2 * I, K).contiguous()
    yg = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=DEV)
    act = torch.empty(M, I, dtype=torch.bfloat16, device=DEV)
    _C.check(lib.cx_gemm_bf16_swiglu(x.data_ptr(), wi.data_ptr(), yg.data_ptr(), act.data_ptr(), M, I, K, K, K, 2 * I, I,
                                     S()))
    y_ref = (x.float() @ w11.float().T).to(torch.bfloat16)
    g_ref = (x.float() @ w12.float().T).to(torch.bfloat16)
    v = yg.view(M, I // 32, 2, 32)
    assert rel_err(v[:, :, 0].reshape(M, I).float(), y_ref.float()) < 4e-3
    assert rel_err(v[:, :, 1].reshape(M, I).float(), g_ref.float()) < 4e-3
    # act must be exactly silu(gate)*y of the STORED bf16 pair (what backward will differentiate)
    yy, gg = v[:, :, 0].reshape(M, I).float(), v[:, :, 1].reshape(M, I).float()
    want = torch.nn.functional.silu(gg) * yy
    assert rel_err(act.float(), want) < 3e-3
    act2 = torch.empty_like(act)
    _C.check(lib.cx_gemm_bf16_swiglu(x.data_ptr(), wi.data_ptr(), None, act2.data_ptr(), M, I, K, K, K, 2 * I, I, S()))
    assert torch.equal(act, act2), "no-grad variant (no pre-activation store) must give identical activations"
    # interleaved-layout standalone ops agree with the fused epilogue
    act3 = torch.empty_like(act)
    _C.check(lib.cx_swiglu_fwd(yg.data_ptr(), act3.data_ptr(), M, I, 1, S()))
    assert rel_err(act3.float(), act.float()) < 2e-3
    _gemm_lib_reset()


@pytest.mark.parametrize("variant", [2, 5, 6])
@pytest.mark.parametrize("T,O,I", [(8192, 768, 3072), (8192, 2304, 768), (1000, 256, 128), (130, 1024, 256)])
def test_wgrad_natural_layout_tn(T, O, I, variant):
    """G += dY^T A straight from the (T,features) row-major operands (ds_read_b64_tr_b16 fragments), vs torch."""
    lib = _gemm_lib(variant)
    if variant == 6 and I % 256:
        assert lib.cx_gemm_bf16_tn_accum(None, None, None, None, 0, T, O, I, O, I, S()) == -1   # product: CX_ERR_SHAPE
        return
    Tp = (T + 63) // 64 * 64
    dy = torch.zeros(Tp, O, dtype=torch.bfloat16, device=DEV)
    a = torch.zeros(Tp, I, dtype=torch.bfloat16, device=DEV)
    dy[:T] = bf(_randn(T, O, seed=82, std=0.1))
    a[:T] = bf(_randn(T, I, seed=83))
    ws = torch.empty(max(2 * O * I, 16 * 768 * 768), device=DEV)
    g = torch.ones(O, I, device=DEV)
    for _ in range(2):
        _C.check(lib.cx_gemm_bf16_tn_accum(dy.data_ptr(), a.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(), T, O, I,
                                           O, I, S()))
    ref = dy.float().T @ a.float()
    e = rel_err(g - 1.0, 2 * ref)
    report("wgrad_tn", T=T, O=O, I=I, e=e)
    _gemm_lib_reset()
    assert e < 1e-5


def test_wgrad_shape_accum_deterministic():
    """wgrad form at the BASELINE chunk: dW(768,3072) += dY^T act over 8192 tokens; bit-identical across runs."""
    T, O, I = 8192, 768, 3072
    dyT, actT = bf(_randn(O, T, seed=80, std=0.1)), bf(_randn(I, T, seed=81))
    ws = torch.empty(16 * O * O, device=DEV)
    outs = []
    for _ in range(2):
        g = torch.zeros(O, I, device=DEV)
        _C.check(L().cx_gemm_bf16_nt_accum(dyT.data_ptr(), actT.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(), O,
                                           I, T, T, T, S()))
        outs.append(g)
    assert torch.equal(outs[0], outs[1]), "fixed-order split-K reduction must be deterministic"
    assert rel_err(outs[0], dyT.float() @ actT.float().T) < 1e-5


def test_gemm_linearity_full_size():
    """Size-independent property at the BASELINE chunk shape: gemm(x1 + x2) == gemm(x1) + gemm(x2) in fp32 out."""
    M, N, K = 8192, 6144, 768
    x1, x2 = bf(_randn(M, K, seed=4)), bf(_randn(M, K, seed=5))
    x1 = (x1.float() * 0.5).to(torch.bfloat16)  # keep x1+x2 exactly representable: halves and sums of bf16
    x2 = (x2.float() * 0.5).to(torch.bfloat16)
    xs = (x1.float() + x2.float())
    exact = xs.to(torch.bfloat16).float().equal(xs)
    w = bf(_randn(N, K, seed=6, std=0.05))
    a = gemm(x1, w, out_mode=1) + gemm(x2, w, out_mode=1)
    if exact:
        b = gemm(xs.to(torch.bfloat16), w, out_mode=1)
        assert rel_err(a, b) < 1e-5
    assert torch.isfinite(a).all()


def test_transpose_and_casts():
    x = bf(_randn(1000, 768, seed=7))
    out = torch.full((768, 1024), 7.0, dtype=torch.bfloat16, device=DEV)
    _C.check(L().cx_transpose_bf16(x.data_ptr(), out.data_ptr(), 1000, 768, 768, 1024, 1024, S()))
    assert torch.equal(out[:, :1000], x.T)
    assert torch.count_nonzero(out[:, 1000:]) == 0, "token padding must be zero-filled"
    w = _randn(300, 200, seed=8)
    o16 = torch.empty(300 * 200, dtype=torch.bfloat16, device=DEV)
    _C.check(L().cx_cast_f32_to_bf16(w.data_ptr(), o16.data_ptr(), w.numel(), S()))
    assert torch.equal(o16.view(300, 200), w.to(torch.bfloat16)), "RNE cast must match torch bit-for-bit"
    ot = torch.empty(200, 300, dtype=torch.bfloat16, device=DEV)
    _C.check(L().cx_cast_transpose_f32_to_bf16(w.data_ptr(), ot.data_ptr(), 300, 200, S()))
    assert torch.equal(ot, w.T.to(torch.bfloat16))
    # the batched form (one launch for every weight shadow of an optimizer step): a device table of CxCastJob
    mats = [_randn(300, 200, seed=8), _randn(64, 768, seed=18), _randn(770, 65, seed=19)]
    outs = [torch.full((m.shape[1], m.shape[0]), 7.0, dtype=torch.bfloat16, device=DEV) for m in mats]
    tab = np.zeros(3, dtype=np.dtype([("in", "u8"), ("out", "u8"), ("rows", "i4"), ("cols", "i4")]))
    for i, (m, o) in enumerate(zip(mats, outs)):
        tab[i] = (m.data_ptr(), o.data_ptr(), m.shape[0], m.shape[1])
    dev_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(DEV)
    tiles = max(((m.shape[0] + 63) // 64) * ((m.shape[1] + 63) // 64) for m in mats)
    _C.check(L().cx_cast_transpose_f32_to_bf16_batched(dev_tab.data_ptr(), 3, tiles, S()))
    for m, o in zip(mats, outs):
        assert torch.equal(o, m.T.to(torch.bfloat16))
    back = torch.empty(300 * 200, dtype=torch.float32, device=DEV)
    _C.check(L().cx_cast_bf16_to_f32(o16.data_ptr(), back.data_ptr(), w.numel(), S()))
    assert torch.equal(back, o16.float())
    f = _randn(130, 70, seed=9)
    ft = torch.empty(70, 130, device=DEV)
    _C.check(L().cx_transpose_f32(f.data_ptr(), ft.data_ptr(), 130, 70, 70, 130, S()))
    assert torch.equal(ft, f.T)


# ------------------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("d", [256, 768, 1024])
@pytest.mark.parametrize("rows", [1, 1003])
def test_layernorm_fwd_bwd(d, rows):
    x0, res = bf(_randn(rows, d, seed=10)), bf(_randn(rows, d, seed=11))
    g, b = 1 + _randn(d, seed=12, std=0.1), _randn(d, seed=13, std=0.1)
    out = torch.empty_like(x0)
    z = torch.empty_like(x0)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    _C.check(L().cx_layernorm_fwd(x0.data_ptr(), res.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(),
                                  z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, d, 1e-12, S()))
    zr = (x0.float() + res.float()).requires_grad_()
    gr, br = g.clone().requires_grad_(), b.clone().requires_grad_()
    ref = torch.nn.functional.layer_norm(zr, (d,), gr, br, 1e-12)
    assert max_err(out.float(), ref) < 0.04, "bf16 output of O(1..4) values: 1 ulp = 2^-7"
    assert rel_err(out.float(), ref) < 4e-3
    assert torch.equal(z, zr.detach().to(torch.bfloat16))
    assert max_err(mean, zr.mean(-1)) < 1e-5
    da, db_ = bf(_randn(rows, d, seed=14)), bf(_randn(rows, d, seed=15))
    ref.backward(da.float() + db_.float())
    dz = torch.empty_like(x0)
    dg, dbeta = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    _C.check(L().cx_layernorm_bwd(da.data_ptr(), db_.data_ptr(), z.data_ptr(), g.data_ptr(), mean.data_ptr(),
                                  rstd.data_ptr(), None, dz.data_ptr(), dg.data_ptr(), dbeta.data_ptr(), None, 0, rows,
                                  d, S()))
    # two-stage (workspace) parameter-gradient reduction: same numbers, accumulates on top of existing values
    ws = torch.empty(1024 * d, device=DEV)
    dz2, dg2, db2 = torch.empty_like(x0), torch.ones(d, device=DEV), torch.ones(d, device=DEV)
    _C.check(L().cx_layernorm_bwd(da.data_ptr(), db_.data_ptr(), z.data_ptr(), g.data_ptr(), mean.data_ptr(),
                                  rstd.data_ptr(), None, dz2.data_ptr(), dg2.data_ptr(), db2.data_ptr(), ws.data_ptr(),
                                  ws.numel(), rows, d, S()))
    assert torch.equal(dz, dz2) and rel_err(dg2 - 1, dg) < 1e-5 and rel_err(db2 - 1, dbeta) < 1e-5
    # the same backward returning the column sums of its dz (bias gradient of the Linear in front), with a dz_extra term
    ex = bf(_randn(rows, d, seed=16))
    dz4, dg4, db4, cs4 = torch.empty_like(x0), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV), torch.ones(d, device=DEV)
    _C.check(L().cx_layernorm_bwd_colsum(da.data_ptr(), db_.data_ptr(), z.data_ptr(), g.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), ex.data_ptr(), dz4.data_ptr(), dg4.data_ptr(), db4.data_ptr(),
                                         cs4.data_ptr(), ws.data_ptr(), ws.numel(), rows, d, S()))
    assert rel_err(dg4, dg) < 1e-5 and rel_err(db4, dbeta) < 1e-5
    assert rel_err(dz4.float(), (dz.float() + ex.float())) < 8e-3 and rel_err(cs4 - 1, dz4.float().sum(0)) < 1e-5
    e_dz, e_dg, e_db = rel_err(dz.float(), zr.grad), rel_err(dg, gr.grad), rel_err(dbeta, br.grad)
    report("layernorm", d=d, rows=rows, e_dz=e_dz, e_dg=e_dg, e_db=e_db)
    assert e_dz < 8e-3, "dz is stored in bf16 and xhat is rebuilt from bf16 z"
    assert e_dg < 5e-3 and e_db < 1e-4


def test_embed_ln_fwd_bwd():
    V, d, B, Sq = 500, 768, 5, 24
    for use_pos in (False, True):
        word, type_e, pos_e = _randn(V, d, seed=20, std=0.5), _randn(2, d, seed=21, std=0.5), _randn(64, d, seed=22, std=0.5)
        g, b = 1 + _randn(d, seed=23, std=0.1), _randn(d, seed=24, std=0.1)
        gen = torch.Generator().manual_seed(25)
        ids = torch.randint(0, V, (B, Sq), generator=gen).to(DEV)
        lens = [24, 3, 17, 24, 9]
        idx = torch.cat([torch.arange(l) + bb * Sq for bb, l in enumerate(lens)]).to(torch.int32).to(DEV)
        T = idx.numel()
        out = torch.empty(T, d, dtype=torch.bfloat16, device=DEV)
        mean, rstd = torch.empty(T, device=DEV), torch.empty(T, device=DEV)
        pe = pos_e if use_pos else None
        _C.check(L().cx_embed_ln_fwd(ids.data_ptr(), idx.data_ptr(), word.data_ptr(), type_e.data_ptr(), _C.ptr(pe),
                                     g.data_ptr(), b.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), T,
                                     Sq, d, 1e-12, S()))
        wr, tr, pr = word.clone().requires_grad_(), type_e.clone().requires_grad_(), pos_e.clone().requires_grad_()
        gr, br = g.clone().requires_grad_(), b.clone().requires_grad_()
        flat = ids.flatten()[idx.long()]
        z = wr[flat] + tr[0] + (pr[(idx.long() % Sq)] if use_pos else 0)
        ref = torch.nn.functional.layer_norm(z, (d,), gr, br, 1e-12)
        assert rel_err(out.float(), ref) < 4e-3
        da = bf(_randn(T, d, seed=26))
        ref.backward(da.float())
        dw, dt, dp = torch.zeros_like(word), torch.zeros(d, device=DEV), torch.zeros_like(pos_e)
        dg, dbt = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        pad = int(flat[0])  # pretend the first token's id is the padding index: its row must get no gradient
        _C.check(L().cx_embed_ln_bwd(da.data_ptr(), None, ids.data_ptr(), idx.data_ptr(), word.data_ptr(),
                                     type_e.data_ptr(), _C.ptr(pe), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                     dw.data_ptr(), dt.data_ptr(), dp.data_ptr() if use_pos else None, dg.data_ptr(),
                                     dbt.data_ptr(), T, Sq, d, pad, S()))
        want_dw = wr.grad.clone()
        want_dw[pad] = 0
        assert rel_err(dw, want_dw) < 1e-4 and torch.count_nonzero(dw[pad]) == 0
        assert rel_err(dt, tr.grad[0]) < 1e-4
        assert rel_err(dg, gr.grad) < 1e-4 and rel_err(dbt, br.grad) < 1e-4
        if use_pos:
            assert rel_err(dp, pr.grad) < 1e-4
        # the sorted (atomics-free) word-row reduction: the same fp32 numbers (fp32 row gradients in the scratch since round 3),
        # bit-identical from run to run, padding row untouched, += semantics into dword
        sids, perm = torch.sort(flat.to(torch.int32), stable=True)
        perm = perm.to(torch.int32)
        runs = []
        for _ in range(2):
            dw2 = torch.ones_like(word)
            dt2, dg2, db2 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
            dp2 = torch.zeros_like(pos_e)
            scratch = torch.empty(T, d, dtype=torch.float32, device=DEV)
            _C.check(L().cx_embed_ln_bwd_sorted(da.data_ptr(), None, ids.data_ptr(), idx.data_ptr(), word.data_ptr(),
                                                type_e.data_ptr(), _C.ptr(pe), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                dw2.data_ptr(), dt2.data_ptr(), dp2.data_ptr() if use_pos else None,
                                                dg2.data_ptr(), db2.data_ptr(), T, Sq, d, pad, V, sids.data_ptr(),
                                                perm.data_ptr(), scratch.data_ptr(), S()))
            runs.append(dw2.clone())
        assert torch.equal(runs[0], runs[1])
        assert rel_err(runs[0] - 1, want_dw) < 1e-4 and torch.all(runs[0][pad] == 1)
        assert rel_err(dt2, tr.grad[0]) < 1e-4 and rel_err(dg2, gr.grad) < 1e-4


# ------------------------------------------------------------------------------------------- activations and bias
def test_swiglu_gelu_biasgrad():
    T, I = 777, 3072
    yg = bf(_randn(T, 2 * I, seed=30))
    act = torch.empty(T, I, dtype=torch.bfloat16, device=DEV)
    _C.check(L().cx_swiglu_fwd(yg.data_ptr(), act.data_ptr(), T, I, 0, S()))
    ygr = yg.float().requires_grad_()
    ref = torch.nn.functional.silu(ygr[:, I:]) * ygr[:, :I]
    assert torch.equal(act, ref.to(torch.bfloat16)) or rel_err(act.float(), ref) < 3e-3
    d = bf(_randn(T, I, seed=31))
    ref.backward(d.float())
    dyg = torch.empty_like(yg)
    _C.check(L().cx_swiglu_bwd(d.data_ptr(), yg.data_ptr(), dyg.data_ptr(), T, I, 0, S()))
    assert rel_err(dyg.float(), ygr.grad) < 4e-3
    pre, bias = bf(_randn(T, I, seed=32)), _randn(I, seed=33, std=0.2)
    _C.check(L().cx_bias_gelu_fwd(pre.data_ptr(), bias.data_ptr(), act.data_ptr(), T, I, S()))
    pr = pre.float().requires_grad_()
    ref = torch.nn.functional.gelu(pr + bias)
    assert rel_err(act.float(), ref) < 3e-3
    ref.backward(d.float())
    dpre = torch.empty_like(pre)
    _C.check(L().cx_bias_gelu_bwd(d.data_ptr(), pre.data_ptr(), bias.data_ptr(), dpre.data_ptr(), T, I, S()))
    assert rel_err(dpre.float(), pr.grad) < 4e-3
    db = torch.zeros(I, device=DEV)
    _C.check(L().cx_bias_grad(d.data_ptr(), db.data_ptr(), T, I, I, S()))
    assert rel_err(db, d.float().sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------------------- attention
def _attn_ref(qkv, lens, cos, sin, scale):
    """fp32 torch reference on the packed varlen tensor; returns out (T,H,64) and autograd handles."""
    T, _, H, D = qkv.shape
    outs, t0 = [], 0
    for l in lens:
        x = qkv[t0:t0 + l]
        q, k, v = x[:, 0], x[:, 1], x[:, 2]
        if cos is not None:
            c, s = cos[:l, None, :], sin[:l, None, :]

            def rot(u):
                u1, u2 = u[..., :32], u[..., 32:]
                return torch.cat([u1 * c - u2 * s, u2 * c + u1 * s], -1)

            # the fused kernel rounds the rotated q/k to bf16 before the MFMA, exactly like the reference's rotary op
            q, k = rot(q), rot(k)
        sc = torch.einsum("qhd,khd->hqk", q, k) * scale
        outs.append(torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), v))
        t0 += l
    return torch.cat(outs, 0)


@pytest.mark.parametrize("lens", [[300, 129, 64], [2048, 1531], [128, 5]])
def test_attention_prerotated_path_equals_rotate_on_load(lens):
    """Long sequences: q and k rotated in place once (cx_rotary_qkv_inplace), forward without tables, backward through
    cx_attn_varlen_bwd_prerotated -- against the kernels that rotate at every load (same rounding points: the rotated rows
    are bf16 in both), and against the fp32 reference."""
    H, D = 3, 64
    T, B, mx = sum(lens), len(lens), max(lens)
    qkv = bf(_randn(T, 3, H, D, seed=42))
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    inv = 1.0 / (1000.0 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(max(512, mx)).float(), inv)
    cos, sin = torch.cos(fr).to(DEV).contiguous(), torch.sin(fr).to(DEV).contiguous()
    scale = 1 / math.sqrt(D)
    do = bf(_randn(T, H, D, seed=43))

    def run(prerot):
        x = qkv.clone()
        out = torch.empty(T, H, D, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(H, T, device=DEV)
        dqkv = torch.full_like(qkv, float("nan"))
        delta = torch.empty(H, T, device=DEV)
        if prerot:
            _C.check(L().cx_rotary_qkv_inplace(x.data_ptr(), cu.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, H, T, mx, 1, S()))
            _C.check(L().cx_attn_varlen_fwd(x.data_ptr(), cu.data_ptr(), None, None, out.data_ptr(), lse.data_ptr(), B, H, T, mx,
                                            scale, S()))
            _C.check(L().cx_attn_varlen_bwd_prerotated(do.data_ptr(), x.data_ptr(), out.data_ptr(), lse.data_ptr(), cu.data_ptr(),
                                                       cos.data_ptr(), sin.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, H, T,
                                                       mx, scale, S()))
        else:
            _C.check(L().cx_attn_varlen_fwd(x.data_ptr(), cu.data_ptr(), cos.data_ptr(), sin.data_ptr(), out.data_ptr(),
                                            lse.data_ptr(), B, H, T, mx, scale, S()))
            _C.check(L().cx_attn_varlen_bwd(do.data_ptr(), x.data_ptr(), out.data_ptr(), lse.data_ptr(), cu.data_ptr(),
                                            cos.data_ptr(), sin.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, H, T, mx, scale, S()))
        return out, dqkv

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert torch.isfinite(g1.float()).all()
    qr = qkv.float().requires_grad_()
    ref = _attn_ref(qr, lens, cos, sin, scale)
    ref.backward(do.float())
    e_o, e_g = rel_err(o1.float(), ref), rel_err(g1.float(), qr.grad)
    report("attention_prerotated", lens=str(lens), e_out=e_o, e_dqkv=e_g, vs_onload_out=rel_err(o1.float(), o0.float()),
           vs_onload_grad=rel_err(g1.float(), g0.float()))
    assert e_o < 8e-3 and e_g < 1.5e-2
    assert rel_err(o1.float(), o0.float()) < 4e-3 and rel_err(g1.float(), g0.float()) < 8e-3


@pytest.mark.parametrize("rotary", [True, False])
@pytest.mark.parametrize("lens", [[128, 64, 100, 1], [197], [300, 129, 64], [128] * 8, [2048, 1531],
                                  [256, 129, 225, 224, 96, 1],    # 128 < max <= 256: the single-pass K / V-resident kernels (round 6)
                                  [197, 197, 197, 197, 197], [255, 130, 2, 160, 33, 193]])
def test_attention_fwd_bwd(rotary, lens):
    H, D = 3, 64
    T, B, mx = sum(lens), len(lens), max(lens)
    qkv = bf(_randn(T, 3, H, D, seed=40))
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    cos = sin = None
    if rotary:
        inv = 1.0 / (1000.0 ** (torch.arange(0, D, 2).float() / D))
        fr = torch.outer(torch.arange(max(512, mx)).float(), inv)
        cos, sin = torch.cos(fr).to(DEV).contiguous(), torch.sin(fr).to(DEV).contiguous()
    scale = 1 / math.sqrt(D)
    out = torch.empty(T, H, D, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(H, T, device=DEV)
    _C.check(L().cx_attn_varlen_fwd(qkv.data_ptr(), cu.data_ptr(), _C.ptr(cos), _C.ptr(sin), out.data_ptr(),
                                    lse.data_ptr(), B, H, T, mx, scale, S()))
    qr = qkv.float().requires_grad_()
    ref = _attn_ref(qr, lens, cos, sin, scale)
    e_out = rel_err(out.float(), ref)
    do = bf(_randn(T, H, D, seed=41))
    ref.backward(do.float())
    dqkv = torch.full_like(qkv, float("nan"))
    delta = torch.empty(H, T, device=DEV)
    _C.check(L().cx_attn_varlen_bwd(do.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), cu.data_ptr(),
                                    _C.ptr(cos), _C.ptr(sin), delta.data_ptr(), dqkv.data_ptr(), B, H, T, mx, scale,
                                    S()))
    e_dq = rel_err(dqkv[:, 0].float(), qr.grad[:, 0])
    e_dk = rel_err(dqkv[:, 1].float(), qr.grad[:, 1])
    e_dv = rel_err(dqkv[:, 2].float(), qr.grad[:, 2])
    report("attention", rotary=rotary, lens=str(lens), e_out=e_out, e_dq=e_dq, e_dk=e_dk, e_dv=e_dv)
    assert torch.isfinite(dqkv.float()).all(), "every dqkv element must be written"
    # bf16 P/dS operands (2^-9) + bf16 rotated q/k: a few 1e-3 relative in the Frobenius norm
    assert e_out < 6e-3
    assert e_dq < 1.5e-2 and e_dk < 1.5e-2 and e_dv < 1.0e-2


def test_rotary_standalone_roundtrip():
    lens, H = [33, 128, 7], 2
    T = sum(lens)
    qkv = bf(_randn(T, 3, H, 64, seed=42))
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    inv = 1.0 / (1000.0 ** (torch.arange(0, 64, 2).float() / 64))
    fr = torch.outer(torch.arange(128).float(), inv)
    cos, sin = torch.cos(fr).to(DEV).contiguous(), torch.sin(fr).to(DEV).contiguous()
    work = qkv.clone()
    _C.check(L().cx_rotary_qkv_inplace(work.data_ptr(), cu.data_ptr(), cos.data_ptr(), sin.data_ptr(), 3, H, T, 128,
                                       1, S()))
    assert torch.equal(work[:, 2], qkv[:, 2]), "v must be untouched"
    pos = torch.cat([torch.arange(l) for l in lens]).to(DEV)
    c, s = cos[pos][:, None, :], sin[pos][:, None, :]
    q = qkv[:, 0].float()
    want = torch.cat([q[..., :32] * c - q[..., 32:] * s, q[..., 32:] * c + q[..., :32] * s], -1)
    # fp32 rotation (the compiler may contract mul+sub into fma) then one bf16 rounding: at most 1 bf16 ulp apart
    assert rel_err(work[:, 0].float(), want) < 3e-3 and max_err(work[:, 0].float(), want) < 0.04
    _C.check(L().cx_rotary_qkv_inplace(work.data_ptr(), cu.data_ptr(), cos.data_ptr(), sin.data_ptr(), 3, H, T, 128,
                                       -1, S()))
    assert rel_err(work.float(), qkv.float()) < 6e-3  # rotate then un-rotate, two bf16 roundings


# --------------------------------------------------------------------------------------------------------- pooling
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("normalize", [1, 0])
def test_pool_normalize(mode, normalize):
    lens, d = [128, 5, 77, 1], 768
    T, B = sum(lens), len(lens)
    h = bf(_randn(T, d, seed=50))
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    emb, nrm = torch.empty(B, d, device=DEV), torch.empty(B, device=DEV)
    _C.check(L().cx_pool_normalize_fwd(h.data_ptr(), cu.data_ptr(), emb.data_ptr(), nrm.data_ptr(), B, d, mode,
                                       normalize, S()))
    hr = h.float().requires_grad_()
    parts, t0 = [], 0
    for l in lens:
        parts.append(hr[t0] if mode == 1 else hr[t0:t0 + l].mean(0))
        t0 += l
    pooled = torch.stack(parts)
    ref = torch.nn.functional.normalize(pooled, dim=-1) if normalize else pooled
    assert max_err(emb, ref) < 2e-6 * (1 if normalize else 50)
    gup = _randn(B, d, seed=51)
    ref.backward(gup)
    dh = torch.full_like(h, float("nan"))
    _C.check(L().cx_pool_normalize_bwd(gup.data_ptr(), emb.data_ptr(), nrm.data_ptr(), cu.data_ptr(), dh.data_ptr(), B,
                                       d, mode, normalize, S()))
    assert torch.isfinite(dh.float()).all()
    assert rel_err(dh.float(), hr.grad) < 4e-3


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("normalize", [1, 0])
def test_layernorm_bwd_pooled_keeps_the_pooled_gradient_in_fp32(mode, normalize):
    """cx_layernorm_bwd_pooled = pooling / normalisation backward + LayerNorm backward of the last LayerNorm in one kernel
    (round 3): against torch fp32 autograd through LN -> pool -> normalize, and against the two-kernel
    route it replaces -- whose bf16 copy of dout is what the parameter gradients lose."""
    lens, d = [128, 5, 77, 1, 64, 0, 33], 768
    T, B = sum(lens), len(lens)
    z = bf(_randn(T, d, seed=60))
    g, b = 1 + _randn(d, seed=61, std=0.1), _randn(d, seed=62, std=0.1)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    out = torch.empty_like(z)
    mean, rstd = torch.empty(T, device=DEV), torch.empty(T, device=DEV)
    _C.check(L().cx_layernorm_fwd(z.data_ptr(), None, g.data_ptr(), b.data_ptr(), out.data_ptr(), None, mean.data_ptr(),
                                  rstd.data_ptr(), T, d, 1e-12, S()))
    emb, nrm = torch.zeros(B, d, device=DEV), torch.ones(B, device=DEV)
    _C.check(L().cx_pool_normalize_fwd(out.data_ptr(), cu.data_ptr(), emb.data_ptr(), nrm.data_ptr(), B, d, mode, normalize, S()))
    gup = _randn(B, d, seed=63)
    # torch fp32 reference on the same bf16 z (the final hidden states enter the pooling in bf16 on both sides)
    zr, gr, br = z.float().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    ln = torch.nn.functional.layer_norm(zr, (d,), gr, br, 1e-12)
    h = ln + (ln.to(torch.bfloat16).float() - ln).detach()   # value: the bf16 hidden states the pooling read; gradient: identity
    parts, t0, keep = [], 0, []
    for i, l in enumerate(lens):
        if l > 0:
            parts.append(h[t0] if mode == 1 else h[t0:t0 + l].mean(0))
            keep.append(i)
        t0 += l
    pooled = torch.stack(parts)
    ref = torch.nn.functional.normalize(pooled, dim=-1) if normalize else pooled
    ref.backward(gup[keep])
    dz = torch.full_like(z, float("nan"))
    dg, dbeta = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    ws = torch.empty(1024 * d, device=DEV)
    _C.check(L().cx_layernorm_bwd_pooled(gup.data_ptr(), emb.data_ptr(), nrm.data_ptr(), cu.data_ptr(), B, mode, normalize,
                                         z.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dz.data_ptr(),
                                         dg.data_ptr(), dbeta.data_ptr(), None, ws.data_ptr(), ws.numel(), T, d, S()))
    assert torch.isfinite(dz.float()).all()
    e_dz, e_dg, e_db = rel_err(dz.float(), zr.grad), rel_err(dg, gr.grad), rel_err(dbeta, br.grad)
    # the route it replaces: bf16 dout in HBM, then the plain LayerNorm backward
    dh = torch.zeros_like(z)
    _C.check(L().cx_pool_normalize_bwd(gup.data_ptr(), emb.data_ptr(), nrm.data_ptr(), cu.data_ptr(), dh.data_ptr(), B, d, mode,
                                       normalize, S()))
    dz2, dg2, db2 = torch.empty_like(z), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    _C.check(L().cx_layernorm_bwd(dh.data_ptr(), None, z.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None,
                                  dz2.data_ptr(), dg2.data_ptr(), db2.data_ptr(), ws.data_ptr(), ws.numel(), T, d, S()))
    e_db2 = rel_err(db2, br.grad)
    report("layernorm_pooled", mode=mode, normalize=normalize, e_dz=e_dz, e_dg=e_dg, e_db=e_db, e_db_two_kernels=e_db2)
    assert e_dz < 8e-3, "dz is stored in bf16"
    assert e_dg < 2e-3 and e_db < 1e-5, "parameter gradients: fp32 dout, fp32 sums"
    assert e_db <= e_db2 + 1e-7
    # deterministic, and accumulating
    # ... and, with a column-sum target, the bias gradient of the Linear in front of the LayerNorm in the same pass
    dz3, dg3, db3, cs3 = torch.empty_like(z), torch.ones(d, device=DEV), torch.ones(d, device=DEV), torch.ones(d, device=DEV)
    _C.check(L().cx_layernorm_bwd_pooled(gup.data_ptr(), emb.data_ptr(), nrm.data_ptr(), cu.data_ptr(), B, mode, normalize,
                                         z.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dz3.data_ptr(),
                                         dg3.data_ptr(), db3.data_ptr(), cs3.data_ptr(), ws.data_ptr(), ws.numel(), T, d, S()))
    assert torch.equal(dz, dz3) and rel_err(dg3 - 1, dg) < 1e-5 and rel_err(db3 - 1, dbeta) < 1e-5
    assert rel_err(cs3 - 1, dz.float().sum(0)) < 1e-5


# --------------------------------------------------------------------------------------------------------- InfoNCE
def _infonce(q, d, labels, scale, coef, want_dscale=False):
    N, dim = q.shape
    G = d.shape[0]
    lib = L()
    ws = torch.empty(lib.cx_infonce_ws_floats(N, G), device=DEV)
    lse, rows = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    _C.check(lib.cx_infonce_fwd(q.data_ptr(), d.data_ptr(), labels.data_ptr(), scale, ws.data_ptr(), lse.data_ptr(),
                                rows.data_ptr(), N, G, dim, q.stride(0), d.stride(0), S()))
    gm, gmt = torch.empty(N, G, device=DEV), torch.empty(G, N, device=DEV)
    qt, dt = torch.empty(dim, N, device=DEV), torch.empty(dim, G, device=DEV)
    dq, dd = torch.empty(N, dim, device=DEV), torch.empty(G, dim, device=DEV)
    dsc = torch.zeros(1, device=DEV)
    _C.check(lib.cx_infonce_bwd(q.data_ptr(), d.data_ptr(), labels.data_ptr(), lse.data_ptr(), scale, coef,
                                gm.data_ptr(), gmt.data_ptr(), qt.data_ptr(), dt.data_ptr(), dq.data_ptr(),
                                dd.data_ptr(), dsc.data_ptr(), N, G, dim, q.stride(0), d.stride(0), S()))
    return lse, rows, dq, dd, gm, dsc


@pytest.mark.parametrize("N,G,dim,neg", [(96, 288, 64, 3), (128, 128, 768, 1), (2048, 16384, 768, 1), (48, 144, 128, 3)])
def test_infonce_vs_torch(N, G, dim, neg):
    q = torch.nn.functional.normalize(_randn(N, dim, seed=60), dim=-1)
    d = torch.nn.functional.normalize(_randn(G, dim, seed=61), dim=-1)
    W = G // (N * neg)
    labels = ((torch.arange(N) + 0 * N) * (G // (N * W))).to(DEV) if W >= 1 else torch.arange(N, device=DEV)
    scale, coef = 50.0, float(W) / N
    lse, rows, dq, dd, gm, dsc = _infonce(q, d, labels, scale, coef)
    qr, dr = q.double().requires_grad_(), d.double().requires_grad_()
    sp = torch.tensor(scale, dtype=torch.float64, device=DEV, requires_grad=True)
    logits = (qr @ dr.T) * sp
    loss = torch.nn.functional.cross_entropy(logits, labels, reduction="sum") * coef
    loss.backward()
    e_loss = abs(float(rows.double().sum() * coef - loss)) / abs(float(loss))
    e_dq, e_dd = rel_err(dq, qr.grad), rel_err(dd, dr.grad)
    e_ds = abs(float(dsc[0]) - float(sp.grad)) / (abs(float(sp.grad)) + 1e-12)
    rowsum = float(gm.sum(1).abs().max())
    report("infonce", N=N, G=G, dim=dim, e_loss=e_loss, e_dq=e_dq, e_dd=e_dd, e_dscale=e_ds, rowsum=rowsum)
    # exact-fp32 MFMA: fp32 round-off only (logits up to 50 -> abs err ~1e-5)
    assert e_loss < 2e-6
    assert max_err(lse, torch.logsumexp(logits, 1)) < 5e-5
    assert e_dq < 2e-5 and e_dd < 2e-5
    assert e_ds < 1e-3
    # size-independent property: every row of (softmax - onehot) sums to zero
    assert rowsum < 1e-5 * coef * scale * 10 + 1e-6


def test_sgemm_nt():
    a, b = _randn(300, 160, seed=70), _randn(200, 160, seed=71)
    c = torch.empty(300, 200, device=DEV)
    _C.check(L().cx_sgemm_nt(a.data_ptr(), b.data_ptr(), c.data_ptr(), 300, 200, 160, 160, 160, 200, S()))
    assert rel_err(c, a.double() @ b.double().T) < 2e-6


@pytest.mark.parametrize("M,N,K", [(8192, 3072, 768), (300, 512, 256), (257, 96, 64)])
def test_gemm_bias_gelu_fused(M, N, K):
    """fc1 + bias + erf-GELU in the GEMM epilogue == GEMM, then bias + GELU in fp32 (one bf16 rounding each)."""

    x, w = bf(_randn(M, K, seed=70)), bf(_randn(N, K, seed=71, std=0.05))
    bias = _randn(N, seed=72)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    act = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    _C.check(L().cx_gemm_bf16_bias_gelu(x.data_ptr(), w.data_ptr(), bias.data_ptr(), pre.data_ptr(), act.data_ptr(), M, N, K,
                                        K, K, N, N, S()))
    ref_pre = x.float() @ w.float().T + bias
    assert rel_err(pre.float(), ref_pre) < 4e-3
    # the activation is defined on the bf16-rounded pre-activation (what the unfused op sees)
    ref_act = torch.nn.functional.gelu(pre.float())
    assert rel_err(act.float(), ref_act) < 4e-3
    assert max_err(act.float(), ref_act) < 2e-2
    act2 = torch.empty_like(act)
    _C.check(L().cx_gemm_bf16_bias_gelu(x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, act2.data_ptr(), M, N, K, K, K, N,
                                        N, S()))
    assert torch.equal(act, act2), "no-grad variant (no pre-activation store) must give identical activations"


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("M,N,K", [(8192, 3072, 768), (1000, 512, 256), (257, 96, 64), (300, 3072, 768)])
def test_gemm_act_bwd_fused_equals_the_two_kernel_backward(M, N, K, act):
    """fc2 dgrad + GELU / quick_gelu backward + fc1 bias gradient in ONE kernel (round 6, sc/layers/mlp.py:30-34 through FusedMLP):
    dPre bit-identical to cx_gemm_bf16_nt followed by cx_bias_act_bwd_colsum(bias = NULL); the bias gradient (column sums of the bf16
    dPre, accumulated INTO dbias) equal to the fp32 sum of that tensor up to summation order; both against an fp32 torch reference."""
    dy, w = bf(_randn(M, K, seed=170)), bf(_randn(N, K, seed=171, std=0.05))
    pre = bf(_randn(M, N, seed=172, std=1.5))
    dact = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    dpre2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    db2 = torch.full((N,), 0.25, device=DEV)
    _C.check(L().cx_gemm_bf16_nt(dy.data_ptr(), w.data_ptr(), dact.data_ptr(), None, M, N, K, K, K, N, 0, 1, 1.0, S()))
    _C.check(L().cx_bias_act_bwd_colsum(dact.data_ptr(), pre.data_ptr(), None, dpre2.data_ptr(), db2.data_ptr(), M, N, act, S()))
    dpre = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    db = torch.full((N,), 0.25, device=DEV)
    nblk = (M + 127) // 128
    ws = torch.full((nblk * N + 5,), float("nan"), device=DEV)
    rc = L().cx_gemm_bf16_act_bwd(dy.data_ptr(), w.data_ptr(), pre.data_ptr(), dpre.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                  M, N, K, K, K, N, N, act, S())
    _C.check(rc)
    assert torch.equal(dpre, dpre2), "fused dPre must be bit-identical to GEMM + standalone activation backward"
    want_db = 0.25 + dpre2.float().sum(0)
    assert rel_err(db, want_db) < 1e-5 and rel_err(db2, want_db) < 1e-5
    # fp32 reference of the op itself
    x = pre.float().requires_grad_()
    y = torch.nn.functional.gelu(x) if act == 0 else x * torch.sigmoid(1.702 * x)
    y.backward(dy.float() @ w.float().T)
    assert rel_err(dpre.float(), x.grad) < 6e-3
    # no bias gradient wanted: no workspace needed, same dPre; a workspace that is too small is declined
    dpre3 = torch.empty_like(dpre)
    _C.check(L().cx_gemm_bf16_act_bwd(dy.data_ptr(), w.data_ptr(), pre.data_ptr(), dpre3.data_ptr(), None, None, 0, M, N, K, K, K, N, N,
                                      act, S()))
    assert torch.equal(dpre3, dpre)
    assert L().cx_gemm_bf16_act_bwd(dy.data_ptr(), w.data_ptr(), pre.data_ptr(), dpre3.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                    nblk * N - 1, M, N, K, K, K, N, N, act, S()) == -1   # CX_ERR_SHAPE


@pytest.mark.parametrize("M,I,K", [(8192, 3072, 768), (1000, 512, 256), (257, 256, 64)])
def test_gemm_swiglu_bwd_fused(M, I, K):
    """fc2 dgrad with the SwiGLU backward in the epilogue == dgrad GEMM (bf16 d(act)) followed by cx_swiglu_bwd."""

    dy = bf(_randn(M, K, seed=80))
    w = bf(_randn(I, K, seed=81, std=0.05))          # transposed fc2 weight: (I, d)
    yg = bf(_randn(M, 2 * I, seed=82))               # interleaved-by-32 [y | g] pre-activations
    got = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=DEV)
    _C.check(L().cx_gemm_bf16_swiglu_bwd(dy.data_ptr(), w.data_ptr(), yg.data_ptr(), got.data_ptr(), M, I, K, K, K, 2 * I,
                                         S()), "cx_gemm_bf16_swiglu_bwd")
    dact = gemm(dy, w, out_mode=0)                   # (M, I) bf16
    want = torch.empty_like(got)
    _C.check(L().cx_swiglu_bwd(dact.data_ptr(), yg.data_ptr(), want.data_ptr(), M, I, 1, S()), "cx_swiglu_bwd")
    # (round 4: the fused epilogue keeps d(act) in fp32, the two-kernel route rounds it to bf16 in between -- one bf16 rounding
    # of an input on top of the two independent output roundings: sqrt(3) x 2^-9 / sqrt(3) ...)
    assert rel_err(got.float(), want.float()) < 4e-3
    # and against plain torch on the bf16 inputs
    v = yg.view(M, I // 32, 2, 32).float()
    y, g = v[:, :, 0].reshape(M, I), v[:, :, 1].reshape(M, I)
    d = dact.float()
    sg = torch.sigmoid(g)
    ref = torch.stack([(g * sg * d).view(M, I // 32, 32), ((sg * (1 + g * (1 - sg))) * d * y).view(M, I // 32, 32)], 2)
    assert rel_err(got.float(), ref.reshape(M, 2 * I)) < 6e-3


# ---- round 3: few output tiles, long K (small batches): split-K over the chip + one fixed-order fold ---------------------
@pytest.mark.parametrize("M,N,K,with_bias,with_res", [(2048, 768, 3072, True, True), (300, 768, 2304, False, True),
                                                       (1000, 264, 1536, True, False)])
def test_gemm_splitk_small_matches_the_one_pass_projection(M, N, K, with_bias, with_res):
    x, w = bf(_randn(M, K, seed=70)), bf(_randn(N, K, seed=71, std=0.05))
    bias = _randn(N, seed=72) if with_bias else None
    res = bf(_randn(M, N, seed=73)) if with_res else None
    ws = torch.empty(8 * M * N, device=DEV)
    got = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    args = (x.data_ptr(), w.data_ptr(), got.data_ptr(), _C.ptr(bias), _C.ptr(res), ws.data_ptr(), ws.numel(), M, N, K, K, K, N, N, S())
    _C.check(L().cx_gemm_bf16_nt_splitk(*args), "cx_gemm_bf16_nt_splitk")
    ref = x.float() @ w.float().T
    if with_bias:
        ref = ref + bias
    ref = ref.to(torch.bfloat16).float()
    if with_res:
        ref = ref + res.float()
    assert torch.isfinite(got.float()).all()
    assert rel_err(got.float(), ref) < 4e-3     # one bf16 rounding of the result (two with the residual)
    if with_res and N % 8 == 0:                 # the fused-epilogue kernel it stands in for: same rounding points
        one = torch.empty_like(got)
        _C.check(L().cx_gemm_bf16_nt_residual(x.data_ptr(), w.data_ptr(), one.data_ptr(), _C.ptr(bias), res.data_ptr(), M, N, K, K,
                                              K, N, N, S()))
        assert rel_err(got.float(), one.float()) < 3e-3
        assert float((got.float() - one.float()).abs().max()) <= 2 ** -6 * float(one.float().abs().max())  # a few 1-ulp flips
    again = torch.empty_like(got)
    _C.check(L().cx_gemm_bf16_nt_splitk(x.data_ptr(), w.data_ptr(), again.data_ptr(), _C.ptr(bias), _C.ptr(res), ws.data_ptr(),
                                        ws.numel(), M, N, K, K, K, N, N, S()))
    assert torch.equal(got, again), "fixed-order reduction: deterministic"


def test_gemm_splitk_small_declines_what_the_one_pass_kernels_do_better():
    """More than 64 output tiles, a short K, or no room for two slabs: CX_ERR_SHAPE (-1), the caller takes the one-pass kernel."""
    x, w = bf(_randn(20000, 3072, seed=70)), bf(_randn(768, 3072, seed=71))
    out = torch.empty(20000, 768, dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(8 * 2048 * 768, device=DEV)
    f = L().cx_gemm_bf16_nt_splitk
    base = (x.data_ptr(), w.data_ptr(), out.data_ptr(), None, None, ws.data_ptr())
    assert f(*base, ws.numel(), 20000, 768, 3072, 3072, 3072, 768, 768, S()) == -1     # 79 x 3 tiles
    assert f(*base, ws.numel(), 2048, 768, 768, 3072, 3072, 768, 768, S()) == -1       # K = 768: 12 K-tiles
    assert f(*base, 7 * 2048 * 768, 2048, 768, 3072, 3072, 3072, 768, 768, S()) == -1  # K = 3072 wants 8 slabs, never fewer
    assert f(*base, ws.numel(), 2048, 768, 3072, 3072, 3072, 768, 768, S()) == 0


# ---- round 3: the gated MLP keeps the gate alone; y is recovered from act = y * silu(gate) inside the derivative ---------
@pytest.mark.parametrize("M,I,K", [(8192, 3072, 768), (300, 512, 256), (257, 96, 64)])
def test_gemm_swiglu_gate_save(M, I, K):
    """cx_gemm_bf16_swiglu_gate: the SAME activation bit for bit as cx_gemm_bf16_swiglu (with or without the save), and G =
    the gate columns of that kernel's interleaved (y, gate) pair, bit for bit, in plain column order."""
    x = bf(_randn(M, K, seed=90))
    w11, w12 = bf(_randn(I, K, seed=91, std=0.05)), bf(_randn(I, K, seed=92, std=0.05))
    wi = torch.stack([w11.view(I // 32, 32, K), w12.view(I // 32, 32, K)], 1).reshape(2 * I, K).contiguous()
    yg = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=DEV)
    act = torch.empty(M, I, dtype=torch.bfloat16, device=DEV)
    _C.check(L().cx_gemm_bf16_swiglu(x.data_ptr(), wi.data_ptr(), yg.data_ptr(), act.data_ptr(), M, I, K, K, K, 2 * I, I, S()))
    g = torch.full((M, I), 7.0, dtype=torch.bfloat16, device=DEV)
    act_g = torch.empty_like(act)
    _C.check(L().cx_gemm_bf16_swiglu_gate(x.data_ptr(), wi.data_ptr(), g.data_ptr(), act_g.data_ptr(), M, I, K, K, K, I, I, S()),
             "cx_gemm_bf16_swiglu_gate")
    assert torch.equal(act_g, act)
    assert torch.equal(g, yg.view(M, I // 32, 2, 32)[:, :, 1].reshape(M, I))
    act_n = torch.empty_like(act)
    _C.check(L().cx_gemm_bf16_swiglu_gate(x.data_ptr(), wi.data_ptr(), None, act_n.data_ptr(), M, I, K, K, K, I, I, S()))
    assert torch.equal(act_n, act), "no-grad variant (nothing saved) must give identical activations"


@pytest.mark.parametrize("M,I,K", [(8192, 3072, 768), (1000, 512, 256), (257, 256, 64)])
def test_gemm_swiglu_bwd_from_act_and_gate(M, I, K):
    """cx_gemm_bf16_swiglu_bwd_gate (fc2 dgrad + SwiGLU backward from the saved (act, gate)) against: the standalone
    cx_swiglu_bwd_gate on the bf16 d(act) of a separate GEMM (same arithmetic: 2e-3), the (y, gate)-based kernel it replaces
    (one more bf16 rounding on the d(gate) half: 8e-3), and fp64 torch on the exact (y, gate) the activation came from."""
    dy = bf(_randn(M, K, seed=80))
    w = bf(_randn(I, K, seed=81, std=0.05))          # transposed fc2 weight: (I, d)
    y, g = bf(_randn(M, I, seed=82)), bf(_randn(M, I, seed=83, std=2.0))
    g[0, :8] = 0.0                                   # a zero gate: act = 0, d(gate) = 0 * (anything finite) = 0, no NaN
    g[1, :8] = -120.0                                # silu underflows: act = 0 -> both gradients 0
    g[2, :8] = 60.0
    act = (torch.nn.functional.silu(g.float()) * y.float()).to(torch.bfloat16)
    got = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=DEV)
    _C.check(L().cx_gemm_bf16_swiglu_bwd_gate(dy.data_ptr(), w.data_ptr(), act.data_ptr(), g.data_ptr(), got.data_ptr(), M, I, K,
                                              K, K, I, 2 * I, S()), "cx_gemm_bf16_swiglu_bwd_gate")
    assert torch.isfinite(got.float()).all()
    dact = gemm(dy, w, out_mode=0)                   # (M, I) bf16
    want = torch.empty_like(got)
    _C.check(L().cx_swiglu_bwd_gate(dact.data_ptr(), act.data_ptr(), g.data_ptr(), want.data_ptr(), M, I, S()), "cx_swiglu_bwd_gate")
    assert rel_err(got.float(), want.float()) < 4e-3   # (the standalone op reads a bf16 d(act), the epilogue keeps it in fp32)
    yg = torch.stack([y.view(M, I // 32, 32), g.view(M, I // 32, 32)], 2).reshape(M, 2 * I).contiguous()
    old = torch.empty_like(got)
    _C.check(L().cx_gemm_bf16_swiglu_bwd(dy.data_ptr(), w.data_ptr(), yg.data_ptr(), old.data_ptr(), M, I, K, K, K, 2 * I, S()))
    gv, ov = got.view(M, I // 32, 2, 32).float(), old.view(M, I // 32, 2, 32).float()
    assert torch.equal(gv[:, :, 0], ov[:, :, 0]), "d y = silu(gate) * d(act) does not involve y: identical"
    e_gate = rel_err(gv[:, :, 1], ov[:, :, 1])
    d64, y64, g64 = dact.double(), y.double(), g.double()
    sg = torch.sigmoid(g64)
    ref_dg = (sg * (1 + g64 * (1 - sg))) * d64 * y64
    # (the planted exact-zero gates are where y is not recoverable -- act = 0 there -- and d(gate) comes out 0 instead of
    # d * y / 2: a probability-zero event for a bf16-rounded fp32 accumulator, left out of the norm below)
    ok = torch.ones(M, I, dtype=torch.bool, device=DEV)
    ok[0, :8] = False
    gn, on = gv[:, :, 1].reshape(M, I).double() * ok, ov[:, :, 1].reshape(M, I).double() * ok
    e_new, e_old = rel_err(gn, ref_dg * ok), rel_err(on, ref_dg * ok)
    report("swiglu_bwd_gate", M=M, I=I, K=K, rel_dgate_vs_yg_kernel=e_gate, rel_dgate_vs_fp64=e_new, yg_kernel_vs_fp64=e_old)
    assert e_gate < 8e-3
    assert e_new < 4e-3 and e_new < 2.0 * e_old + 1e-3   # (one more bf16 rounding than the (y, gate) kernel: sqrt(2) x)
    assert float(got.view(M, I // 32, 2, 32)[0, 0, 1, :8].float().abs().max()) == 0.0   # zero gate -> zero d(gate)


@pytest.mark.parametrize("with_bias", [False, True])
@pytest.mark.parametrize("M,N,K", [(8192, 768, 768), (1000, 768, 3072), (257, 264, 64)])
def test_gemm_residual_fused(M, N, K, with_bias):
    """Projection + residual add in the GEMM epilogue == bf16 GEMM output, then fp32 add, one more bf16 rounding."""

    x, w = bf(_randn(M, K, seed=60)), bf(_randn(N, K, seed=61, std=0.05))
    res = bf(_randn(M, N, seed=62))
    bias = _randn(N, seed=63) if with_bias else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    _C.check(L().cx_gemm_bf16_nt_residual(x.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr() if with_bias else None,
                                          res.data_ptr(), M, N, K, K, K, N, N, S()), "cx_gemm_bf16_nt_residual")
    base = gemm(x, w, bias=bias, out_mode=0)
    want = (base.float() + res.float()).to(torch.bfloat16)
    assert rel_err(out.float(), want.float()) < 1e-3
    assert float((out.float() - want.float()).abs().max()) <= 0.0625  # at most one bf16 ulp of O(4) values


# ---- round 4: the two-workgroups-per-CU kernel (gemm_bf16_v7.hip) against the one-wave-per-SIMD kernel it is routed next to
def _v7_pair(run):
    """Run `run(lib)` with every covered launch forced onto v7, then with v7 off; the dev library carries the switch."""
    lib = LD()
    try:
        lib.cx_gemm_v7_mode(1)
        a = run(lib)
        lib.cx_gemm_v7_mode(0)
        b = run(lib)
    finally:
        lib.cx_gemm_v7_mode(-1)
    return a, b


@pytest.mark.parametrize("M,N,K", [(8192, 768, 768), (8192, 2304, 768), (4100, 768, 3072), (1000, 3072, 128), (257, 128, 64),
                                   (64, 256, 768), (16384 + 77, 6144, 768)])
@pytest.mark.parametrize("with_res,with_bias", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_v7_plain_and_residual_bit_identical_to_v6(M, N, K, with_res, with_bias):
    """Same k-ascending chain of 32x32x16 MFMAs per output element, same rounding points in the epilogue: the 256x128 /
    two-workgroups-per-CU kernel must reproduce the 256x256 / one-wave-per-SIMD kernel bit for bit -- interior tiles, a
    partial last M-panel (4100, 1000, 257, 64, 16461 rows), K from one K-tile pair to 48."""
    x, w = bf(_randn(M, K, seed=70)), bf(_randn(N, K, seed=71, std=0.05))
    res = bf(_randn(M, N, seed=72))
    bias = _randn(N, seed=69) if with_bias else None
    bp = bias.data_ptr() if with_bias else None

    def run(lib):
        out = torch.full((M + 3, N), 7.0, dtype=torch.bfloat16, device=DEV)   # 3 guard rows: nothing may be written past M
        if with_res:
            _C.check(lib.cx_gemm_bf16_nt_residual(x.data_ptr(), w.data_ptr(), out.data_ptr(), bp, res.data_ptr(), M, N, K, K, K, N, N, S()))
        else:
            _C.check(lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), out.data_ptr(), bp, M, N, K, K, K, N, 0, 1, 1.0, S()))
        torch.cuda.synchronize()
        return out

    a, b = _v7_pair(run)
    assert torch.equal(a[M:], torch.full_like(a[M:], 7.0)), "rows past M were written"
    assert torch.equal(a, b)
    ref = x.float() @ w.float().T + (bias if with_bias else 0.0)
    if with_res:
        ref = ref.bfloat16().float() + res.float()
    assert rel_err(a[:M].float(), ref) < 5e-3


@pytest.mark.parametrize("M,I,K", [(8192, 3072, 768), (4100, 512, 256), (257, 64, 64), (70, 3072, 768)])
@pytest.mark.parametrize("save", [True, False])
def test_gemm_v7_swiglu_gate_bit_identical_to_v6(M, I, K, save):
    x = bf(_randn(M, K, seed=73))
    w11, w12 = bf(_randn(I, K, seed=74, std=0.05)), bf(_randn(I, K, seed=75, std=0.05))
    wi = torch.stack([w11.view(I // 32, 32, K), w12.view(I // 32, 32, K)], 1).reshape(2 * I, K).contiguous()

    def run(lib):
        g = torch.full((M + 2, I), 7.0, dtype=torch.bfloat16, device=DEV)
        act = torch.full((M + 2, I), 7.0, dtype=torch.bfloat16, device=DEV)
        _C.check(lib.cx_gemm_bf16_swiglu_gate(x.data_ptr(), wi.data_ptr(), g.data_ptr() if save else None, act.data_ptr(), M, I, K, K, K, I, I, S()))
        torch.cuda.synchronize()
        return g, act

    (g7, a7), (g6, a6) = _v7_pair(run)
    assert torch.equal(a7, a6) and torch.equal(g7, g6)
    assert torch.equal(a7[M:], torch.full_like(a7[M:], 7.0))
    y, gate = x.float() @ w11.float().T, x.float() @ w12.float().T
    want = torch.nn.functional.silu(gate.bfloat16().float()) * y.bfloat16().float()
    assert rel_err(a7[:M].float(), want) < 6e-3
    if save:
        assert rel_err(g7[:M].float(), gate) < 4e-3


@pytest.mark.parametrize("M,I,K", [(8192, 3072, 768), (4100, 512, 256), (257, 256, 64), (70, 3072, 768)])
def test_gemm_v7_swiglu_bwd_gate_bit_identical_to_v6(M, I, K):
    dy = bf(_randn(M, K, seed=76))
    w = bf(_randn(I, K, seed=77, std=0.05))
    y, g = bf(_randn(M, I, seed=78)), bf(_randn(M, I, seed=79, std=2.0))
    g[0, :8] = 0.0
    g[1, :8] = -120.0
    act = (torch.nn.functional.silu(g.float()) * y.float()).to(torch.bfloat16)

    def run(lib):
        out = torch.full((M + 2, 2 * I), 7.0, dtype=torch.bfloat16, device=DEV)
        _C.check(lib.cx_gemm_bf16_swiglu_bwd_gate(dy.data_ptr(), w.data_ptr(), act.data_ptr(), g.data_ptr(), out.data_ptr(), M, I, K, K, K, I, 2 * I, S()))
        torch.cuda.synchronize()
        return out

    a, b = _v7_pair(run)
    assert torch.isfinite(a.float()).all()
    assert torch.equal(a[M:], torch.full_like(a[M:], 7.0))
    assert torch.equal(a, b)


def test_gemm_v7_is_deterministic_and_race_free_at_full_size():
    """The hand-placed waits of the LDS-DMA ring (counted vmcnt, one barrier per K-tile, a private X slot per wave) are
    correct by count, not by luck: the metric's fc2-dgrad + SwiGLU-backward launch five times over, every run bit-identical to
    the first and to the v6 kernel."""
    M, I, K = 32768, 3072, 768
    dy = bf(_randn(M, K, seed=81))
    w = bf(_randn(I, K, seed=82, std=0.05))
    act, g = bf(_randn(M, I, seed=83)), bf(_randn(M, I, seed=84, std=2.0))
    lib = LD()
    outs = []
    try:
        for mode in (1, 1, 1, 1, 1, 0):
            lib.cx_gemm_v7_mode(mode)
            out = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=DEV)
            _C.check(lib.cx_gemm_bf16_swiglu_bwd_gate(dy.data_ptr(), w.data_ptr(), act.data_ptr(), g.data_ptr(), out.data_ptr(), M, I, K, K, K, I, 2 * I, S()))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib.cx_gemm_v7_mode(-1)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


# ---- the plain one-wave-per-SIMD form on its own (gemm_bf16_v6.hip; v7 off: the small shapes would otherwise be routed there) ----
@pytest.mark.parametrize("M,N,K", [(65536, 2304, 768), (65536, 768, 768), (16384, 3072, 768), (8192, 768, 3072), (512, 256, 512),
                                   (256, 256, 6144), (66048, 768, 2304), (2048, 6144, 1024), (300, 264, 64), (256, 256, 64)])
def test_gemm_v6_plain_shapes_against_fp32_and_guard_rows(M, N, K):
    """Many tiles per workgroup (the metric's qkv / out_proj shapes), exactly one tile per workgroup, fewer tiles than workgroups,
    partial last panels, K from one K-tile to 96; EVERY element against the fp32 product of the same bf16 operands (an element
    that landed in the wrong place of the 16 x 16 block map shows as an O(1) error), guard rows behind the output untouched."""
    x, w = bf(_randn(M, K, seed=90)), bf(_randn(N, K, seed=91, std=0.05))
    lib = LD()
    try:
        lib.cx_gemm_v7_mode(0)
        out = torch.full((M + 3, N), 7.0, dtype=torch.bfloat16, device=DEV)
        _C.check(lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, K, K, K, N, 0, 1, 1.0, S()))
        torch.cuda.synchronize()
    finally:
        lib.cx_gemm_v7_mode(-1)
    assert torch.equal(out[M:], torch.full_like(out[M:], 7.0)), "rows past M were written"
    for r0 in range(0, M, 16384):
        ref = x[r0:r0 + 16384].float() @ w.float().T
        got = out[r0:min(r0 + 16384, M)].float()
        assert (got - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-3
        assert rel_err(got, ref) < 5e-3


def test_gemm_v6_strided_output_and_repeatable():
    """Output with a leading dimension wider than N (a column slice of a wider tensor: the stores address rows through ldo) and
    five repeats of the metric-sized qkv launch: the counted waits must hold by construction, every run bit-identical."""
    M, N, K = 65536, 2304, 768
    x, w = bf(_randn(M, K, seed=92)), bf(_randn(N, K, seed=93, std=0.05))
    lib = LD()
    outs = []
    try:
        lib.cx_gemm_v7_mode(0)
        for _ in range(5):
            wide = torch.full((M, N + 256), 7.0, dtype=torch.bfloat16, device=DEV)
            _C.check(lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), wide[:, 128:].data_ptr(), None, M, N, K, K, K, N + 256, 0, 1, 1.0, S()))
            torch.cuda.synchronize()
            outs.append(wide)
    finally:
        lib.cx_gemm_v7_mode(-1)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert torch.equal(outs[0][:, :128], torch.full_like(outs[0][:, :128], 7.0))
    assert torch.equal(outs[0][:, N + 128:], torch.full_like(outs[0][:, N + 128:], 7.0))
    assert rel_err(outs[0][:4096, 128:N + 128].float(), x[:4096].float() @ w.float().T) < 5e-3


# ---- fused optimizer tail (optimizer.hip) vs torch.optim.AdamW + clip_grad_norm_ -----------------------------------
@pytest.mark.parametrize("n,max_norm,wd", [(1 << 20, 1.0, 0.01), (4099, 0.05, 0.0), (3, None, 0.1), (786432 + 2, 1e9, 0.01)])
def test_fused_adamw_clip_matches_torch(n, max_norm, wd):
    from contrastors_amd.optimizer import FusedAdamW

    torch.manual_seed(n)
    base = torch.randn(n + 8, device="cuda")
    p_ref = torch.nn.Parameter(base[:n].clone())
    p_new = torch.nn.Parameter(base[:n].clone())
    extra_ref = torch.nn.Parameter(base[n:n + 8].clone())  # a second tensor: the norm is global over all of them
    extra_new = torch.nn.Parameter(base[n:n + 8].clone())
    ref = torch.optim.AdamW([{"params": [p_ref], "weight_decay": wd}, {"params": [extra_ref], "weight_decay": 0.0}],
                            lr=3e-3, betas=(0.9, 0.98), eps=1e-8)
    new = FusedAdamW([{"params": [p_new], "weight_decay": wd}, {"params": [extra_new], "weight_decay": 0.0}],
                     lr=3e-3, betas=(0.9, 0.98), eps=1e-8)
    sched_ref = torch.optim.lr_scheduler.LambdaLR(ref, lambda s: 1.0 / (1 + s))
    sched_new = torch.optim.lr_scheduler.LambdaLR(new, lambda s: 1.0 / (1 + s))
    for it in range(4):
        g = torch.randn(n + 8, device="cuda") * (0.5 + it)
        p_ref.grad, extra_ref.grad = g[:n].clone(), g[n:].clone()
        p_new.grad, extra_new.grad = g[:n].clone(), g[n:].clone()
        if max_norm is not None:
            total = torch.nn.utils.clip_grad_norm_([p_ref, extra_ref], max_norm)
        ref.step()
        new.step(max_grad_norm=max_norm)
        sched_ref.step()
        sched_new.step()
        if max_norm is not None:
            assert abs(float(new.last_grad_norm) - float(total)) <= 1e-5 * float(total)
        # the fused step must not have touched the gradient (the reference's clip scales it in place; nothing reads it after)
        assert torch.equal(p_new.grad, g[:n])
        assert float((p_new - p_ref).detach().abs().max()) <= 2e-6 * (1 + float(p_ref.detach().abs().max()))
        assert float((extra_new - extra_ref).detach().abs().max()) <= 2e-6 * (1 + float(extra_ref.detach().abs().max()))
    sr, sn = ref.state[p_ref], new.state[p_new]
    assert float(sr["step"]) == float(sn["step"]) == 4
    torch.testing.assert_close(sn["exp_avg"], sr["exp_avg"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(sn["exp_avg_sq"], sr["exp_avg_sq"], rtol=1e-5, atol=1e-9)
    # state dicts are interchangeable with torch.optim.AdamW's (optimizer.pt of save_state / load_state)
    ref2 = torch.optim.AdamW([{"params": [torch.nn.Parameter(p_new.detach().clone())], "weight_decay": wd},
                              {"params": [torch.nn.Parameter(extra_new.detach().clone())], "weight_decay": 0.0}],
                             lr=3e-3, betas=(0.9, 0.98), eps=1e-8)
    ref2.load_state_dict(new.state_dict())
    assert float(ref2.state[ref2.param_groups[0]["params"][0]]["step"]) == 4
