"""Same-box A/B of whole LIBRARIES (the product .so against variants built by scripts/build_variant.py) on isolated launches of
the step's kernels: every library is loaded side by side (ctypes), rounds are interleaved (lib A, lib B, ..., lib A, ...) so that
clock / power drift of the box hits all of them alike, and every variant's output is compared with the base library's (bit-identical
or the max relative difference).  usage:
    python scripts/lib_ab.py [--libs base,hi1,hi2] [--cases swiglu_bwd,attn_bwd] [--chunk 2048] [--rounds 7] [--reps 6]
`base` = contrastors_amd/lib/libcontrastors_hip.so; any other name N = contrastors_amd/lib/variants/libcontrastors_hip_N.so."""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--libs", type=str, default="base")
ap.add_argument("--cases", type=str, default="")
ap.add_argument("--chunk", type=int, default=2048)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--seq", type=int, default=128)
ap.add_argument("--rotary", type=int, default=1, help="0: no rotation tables (pre-rotated long sequences, image towers)")
a = ap.parse_args()


def load(name):
    path = _C.LIB_PATH if name == "base" else _C.LIB_PATH.parent / "variants" / f"libcontrastors_hip_{name}.so"
    h = C.CDLL(str(path))
    for fn, (res, args) in _C._SIGS.items():
        if hasattr(h, fn):
            f = getattr(h, fn)
            f.restype, f.argtypes = res, args
    return h


names = a.libs.split(",")
libs = {n: load(n) for n in names}
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *sh, std=1.0: (torch.randn(*sh, device=dev, generator=g) * std).bfloat16()   # noqa: E731
P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
T, d, I, H, S = a.chunk * 128, 768, 3072, 12, a.seq
x, res = rn(T, d), rn(T, d)
w1, w2t, w2, wo, wqkv = rn(2 * I, d, std=0.05), rn(I, d, std=0.05), rn(d, I, std=0.05), rn(d, d, std=0.05), rn(3 * d, d, std=0.05)
act, gate = rn(T, I), rn(T, I, std=2.0)
x3, wqkv_t = rn(T, 3 * d), rn(d, 3 * d, std=0.05)
x6, w1_t = rn(T, 2 * I), rn(d, 2 * I, std=0.05)
dyg = torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16)
out_d = torch.empty(T, d, device=dev, dtype=torch.bfloat16)
out_3d = torch.empty(T, 3 * d, device=dev, dtype=torch.bfloat16)
out_I = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
gs = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
# attention (S <= 128 single-pass kernels at S = 128; ragged: lengths 64..128)
B = T // S
qkv = rn(T, 3 * H * 64, std=0.5)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=dev)
inv = 1.0 / (1000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
fr = torch.outer(torch.arange(max(S, 128), dtype=torch.float32), inv)
cos, sin = torch.cos(fr).to(dev).contiguous(), torch.sin(fr).to(dev).contiguous()
if not a.rotary:
    cos = sin = None
att_out = torch.empty(T, H * 64, device=dev, dtype=torch.bfloat16)
lse = torch.empty(H * T, device=dev)
dout = rn(T, H * 64)
dqkv = torch.empty_like(qkv)
delta = torch.empty(H * T, device=dev)
lens = torch.randint(S // 2, S + 1, (B,), generator=torch.Generator().manual_seed(1))
cu_r = torch.zeros(B + 1, dtype=torch.int32)
cu_r[1:] = lens.cumsum(0)
cu_r = cu_r.to(dev)
T_r = int(lens.sum())

# name -> (flop, outputs to compare, call(lib))
cases = {
    "swiglu_bwd": (2.0 * T * I * d, [dyg], lambda L: L.cx_gemm_bf16_swiglu_bwd_gate(P(x), P(w2t), P(act), P(gate), P(dyg), T, I, d, d, d, I, 2 * I, s)),
    "swiglu_fwd_save": (2.0 * T * 2 * I * d, [gs, out_I], lambda L: L.cx_gemm_bf16_swiglu_gate(P(x), P(w1), P(gs), P(out_I), T, I, d, d, d, I, I, s)),
    "swiglu_fwd": (2.0 * T * 2 * I * d, [out_I], lambda L: L.cx_gemm_bf16_swiglu_gate(P(x), P(w1), None, P(out_I), T, I, d, d, d, I, I, s)),
    "qkv_fwd": (2.0 * T * 3 * d * d, [out_3d], lambda L: L.cx_gemm_bf16_nt(P(x), P(wqkv), P(out_3d), None, T, 3 * d, d, d, d, 3 * d, 0, 1, 1.0, s)),
    "out_dgrad": (2.0 * T * d * d, [out_d], lambda L: L.cx_gemm_bf16_nt(P(x), P(wo), P(out_d), None, T, d, d, d, d, d, 0, 1, 1.0, s)),
    "out_fwd_res": (2.0 * T * d * d, [out_d], lambda L: L.cx_gemm_bf16_nt_residual(P(x), P(wo), P(out_d), None, P(res), T, d, d, d, d, d, d, s)),
    "fc2_fwd_res": (2.0 * T * I * d, [out_d], lambda L: L.cx_gemm_bf16_nt_residual(P(act), P(w2), P(out_d), None, P(res), T, d, I, I, I, d, d, s)),
    "qkv_dgrad_res": (2.0 * T * 3 * d * d, [out_d], lambda L: L.cx_gemm_bf16_nt_residual(P(x3), P(wqkv_t), P(out_d), None, P(res), T, d, 3 * d, 3 * d, 3 * d, d, d, s)),
    "fc1_dgrad_res": (2.0 * T * 2 * I * d, [out_d], lambda L: L.cx_gemm_bf16_nt_residual(P(x6), P(w1_t), P(out_d), None, P(res), T, d, 2 * I, 2 * I, 2 * I, d, d, s)),
    "attn_fwd": (4.0 * S * S * 64 * B * H, [att_out, lse], lambda L: L.cx_attn_varlen_fwd(P(qkv), P(cu), P(cos), P(sin), P(att_out), P(lse), B, H, T, S, 0.125, s)),
    "attn_bwd": (10.0 * S * S * 64 * B * H, [dqkv], lambda L: L.cx_attn_varlen_bwd(P(dout), P(qkv), P(att_out), P(lse), P(cu), P(cos), P(sin), P(delta), P(dqkv), B, H, T, S, 0.125, s)),
    # (A/B of CX_ATTN_DELTA_IN builds: `delta` is filled beforehand by the base library's general kernels, see below)
    "attn_bwd_dpre": (10.0 * S * S * 64 * B * H, [dqkv], lambda L: L.cx_attn_varlen_bwd(P(dout), P(qkv), P(att_out), P(lse), P(cu), P(cos), P(sin), P(delta), P(dqkv), B, H, T, S, 0.125, s)),
    "attn_bwd_drop": (10.0 * S * S * 64 * B * H, [dqkv], lambda L: L.cx_attn_varlen_dropout_bwd(P(dout), P(qkv), P(att_out), P(lse), P(cu), P(cos), P(sin), P(delta), P(dqkv), B, H, T, S, 0.125, 0.1, 1234, 0, 0, s)),
    "attn_fwd_drop": (4.0 * S * S * 64 * B * H, [att_out, lse], lambda L: L.cx_attn_varlen_dropout_fwd(P(qkv), P(cu), P(cos), P(sin), P(att_out), P(lse), B, H, T, S, 0.125, 0.1, 1234, 0, 0, s)),
    "attn_bwd_ragged": (0.0, [dqkv], lambda L: L.cx_attn_varlen_bwd(P(dout), P(qkv), P(att_out), P(lse), P(cu_r), P(cos), P(sin), P(delta), P(dqkv), B, H, T_r, S, 0.125, s)),
}
want = [c for c in a.cases.split(",") if c] or list(cases)
print(f"# T = {T} token rows, seq {S}; median of {a.rounds} interleaved rounds x {a.reps} launches (us); libs: {names}")
hdr = f"{'case':18s}" + "".join(f"{n + ' us':>12s}{'TF':>8s}" for n in names) + "".join(f"{n + '/base':>12s}{'maxrel':>10s}" for n in names[1:])
print(hdr)
for cname in want:
    fl, outs, call = cases[cname]
    if cname.startswith("attn_bwd"):   # its inputs: a forward of the base library on the same sequences
        libs[names[0]].cx_attn_varlen_fwd(P(qkv), P(cu_r if cname.endswith("ragged") else cu), P(cos), P(sin), P(att_out), P(lse), B, H,
                                           T_r if cname.endswith("ragged") else T, S, 0.125, s)
    if cname == "attn_bwd_dpre":   # delta = rowsum(dO * O) into `delta` (H, T): the general (max_seqlen > 128) path computes it first
        assert libs[names[0]].cx_attn_varlen_bwd(P(dout), P(qkv), P(att_out), P(lse), P(cu), P(cos), P(sin), P(delta), P(dqkv), B, H, T, S + 1, 0.125, s) == 0
        torch.cuda.synchronize()
    ref, diffs = None, {}
    for n in names:
        for o in outs:
            o.zero_()
        rc = call(libs[n])
        assert rc == 0, (cname, n, rc)
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        if ref is None:
            ref = got
        else:
            same = all(torch.equal(x_, y_) for x_, y_ in zip(got, ref))
            rel = max(float((x_.float() - y_.float()).norm() / (y_.float().norm() + 1e-30)) for x_, y_ in zip(got, ref))
            diffs[n] = "bit-ident" if same else f"{rel:.2e}"
    t = {n: [] for n in names}
    for _ in range(a.rounds):
        for n in names:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                call(libs[n])
            e1.record()
            torch.cuda.synchronize()
            t[n].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    med = {n: sorted(v)[len(v) // 2] for n, v in t.items()}
    row = f"{cname:18s}" + "".join(f"{med[n]:12.1f}{(fl / med[n] / 1e6 if fl else 0):8.1f}" for n in names)
    row += "".join(f"{med[n] / med[names[0]]:12.3f}{diffs[n]:>10s}" for n in names[1:])
    print(row, flush=True)
