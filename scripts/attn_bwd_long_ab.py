"""the fused single-owner long-sequence attention backward (attn_bwd_fused_long_kernel, dev library) against the shipped
two-kernel backward (attn_bwd_dq_kernel + attn_bwd_dkv_kernel) on the same inputs: dK / dV expected bit-identical, dQ equal up to the fp32
summation order (both compared with an fp32 torch reference on two sequences), interleaved timing.
usage: python scripts/attn_bwd_long_ab.py [--seq 2048] [--tokens 131072] [--rotary 0|1] [--ragged 0|1] [--pdrop 0.0] [--libs dev,fl2]
`dev` = contrastors_amd/lib/libcontrastors_hip_dev.so; any other name N = lib/variants/libcontrastors_hip_dev_N.so (scripts/build_variant.py)"""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seq", type=int, default=2048)
ap.add_argument("--tokens", type=int, default=131072)
ap.add_argument("--rotary", type=int, default=0, help="1: rotated q / k in qkv, tables un-rotate the gradients (the engine's long-sequence path)")
ap.add_argument("--ragged", type=int, default=0)
ap.add_argument("--pdrop", type=float, default=0.0)
ap.add_argument("--libs", type=str, default="dev")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--timing-only", type=int, default=0)
a = ap.parse_args()


def load(name):
    path = _C.DEV_LIB_PATH if name == "dev" else _C.LIB_PATH.parent / "variants" / f"libcontrastors_hip_dev_{name}.so"
    h = C.CDLL(str(path))
    for fn, (res, args) in {**_C._SIGS, **_C._DEV_SIGS}.items():
        if hasattr(h, fn):
            f = getattr(h, fn)
            f.restype, f.argtypes = res, args
    return h


names = a.libs.split(",")
libs = {n: load(n) for n in names}
base = libs[names[0]]
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
H, S = 12, a.seq
B = max(1, a.tokens // S)
if a.ragged:
    lens = torch.randint(S // 2, S + 1, (B,), generator=torch.Generator().manual_seed(1))
    lens[0] = S
else:
    lens = torch.full((B,), S, dtype=torch.int64)
cu = torch.zeros(B + 1, dtype=torch.int32)
cu[1:] = lens.cumsum(0)
T = int(cu[-1])
cu = cu.to(dev)
P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
qkv = (torch.randn(T, 3 * H * 64, device=dev, generator=g) * 0.5).bfloat16()
dout = torch.randn(T, H * 64, device=dev, generator=g).bfloat16()
cos = sin = None
if a.rotary:
    inv = 1.0 / (1000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
    fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    cos, sin = torch.cos(fr).to(dev).contiguous(), torch.sin(fr).to(dev).contiguous()
out = torch.empty(T, H * 64, device=dev, dtype=torch.bfloat16)
lse = torch.empty(H * T, device=dev)
delta = torch.empty(H * T, device=dev)
scale = 0.125
seed, off, site = 1234, 0, 3
# forward on the (rotated) qkv as it is: no tables at the loads in either backward (the engine's long-sequence path)
if a.pdrop > 0:
    assert base.cx_attn_varlen_dropout_fwd(P(qkv), P(cu), None, None, P(out), P(lse), B, H, T, S, scale, a.pdrop, seed, off, site, s) == 0
else:
    assert base.cx_attn_varlen_fwd(P(qkv), P(cu), None, None, P(out), P(lse), B, H, T, S, scale, s) == 0
torch.cuda.synchronize()


def two_kernel(L, dst):
    if a.pdrop > 0:
        assert not a.rotary
        return L.cx_attn_varlen_dropout_bwd(P(dout), P(qkv), P(out), P(lse), P(cu), None, None, P(delta), P(dst), B, H, T, S, scale, a.pdrop, seed, off, site, s)
    if a.rotary:
        return L.cx_attn_varlen_bwd_prerotated(P(dout), P(qkv), P(out), P(lse), P(cu), P(cos), P(sin), P(delta), P(dst), B, H, T, S, scale, s)
    return L.cx_attn_varlen_bwd(P(dout), P(qkv), P(out), P(lse), P(cu), None, None, P(delta), P(dst), B, H, T, S, scale, s)


ws = torch.empty(int(base.cx_attn_bwd_fused_long_ws_floats(B, H, T)), device=dev)


def fused(L, dst):
    return L.cx_attn_varlen_bwd_fused_long(P(dout), P(qkv), P(out), P(lse), P(cu), P(cos), P(sin), 1 if a.rotary else 0, P(delta), P(dst), P(ws),
                                           B, H, T, S, scale, a.pdrop, seed, off, site, s)


ref = torch.zeros_like(qkv)
assert two_kernel(base, ref) == 0
torch.cuda.synchronize()
print(f"# T = {T} tokens, {B} sequences of <= {S}, H = {H}, rotary (pre-rotated) = {a.rotary}, ragged = {a.ragged}, p_drop = {a.pdrop}; scratch {ws.numel() * 4 / 2**20:.0f} MiB")
# fp32 reference on the first two sequences / two heads (no dropout, no rotation of the gradients)
if a.pdrop == 0 and not a.rotary and not a.timing_only:
    for b in range(min(2, B)):
        t0, t1 = int(cu[b]), int(cu[b + 1])
        x = qkv[t0:t1].float().view(-1, 3, H, 64)
        for h in (0, H - 1):
            q, k, v = (x[:, i, h].clone().requires_grad_(True) for i in range(3))
            o = torch.softmax(q @ k.T * scale, -1) @ v
            o.backward(dout[t0:t1].float().view(-1, H, 64)[:, h])
            want = torch.stack([q.grad, k.grad, v.grad], 1)
            for nm, L in libs.items():
                got = torch.zeros_like(qkv)
                ws.fill_(float("nan"))
                assert fused(L, got) == 0
                torch.cuda.synchronize()
                gt = got[t0:t1].float().view(-1, 3, H, 64)[:, :, h]
                rf = ref[t0:t1].float().view(-1, 3, H, 64)[:, :, h]
                e_f = [float((gt[:, i] - want[:, i]).norm() / want[:, i].norm()) for i in range(3)]
                e_r = [float((rf[:, i] - want[:, i]).norm() / want[:, i].norm()) for i in range(3)]
                print(f"# seq {b} head {h} [{nm}] rel err vs fp32 torch  fused dq/dk/dv = {e_f[0]:.2e} {e_f[1]:.2e} {e_f[2]:.2e}   two-kernel = {e_r[0]:.2e} {e_r[1]:.2e} {e_r[2]:.2e}")
for nm, L in ([] if a.timing_only else libs.items()):
    got = torch.zeros_like(qkv)
    ws.fill_(float("nan"))
    assert fused(L, got) == 0
    torch.cuda.synchronize()
    gv, rv = got.view(T, 3, H * 64), ref.view(T, 3, H * 64)
    rel = lambda i: float((gv[:, i].float() - rv[:, i].float()).norm() / rv[:, i].float().norm())   # noqa: E731
    print(f"# [{nm}] fused vs two-kernel: dK bit-identical {torch.equal(gv[:, 1], rv[:, 1])}, dV bit-identical {torch.equal(gv[:, 2], rv[:, 2])}, "
          f"dQ rel diff {rel(0):.2e} (max abs {float((gv[:, 0].float() - rv[:, 0].float()).abs().max()):.3e}), finite {bool(torch.isfinite(got.float()).all())}")
scr = torch.empty_like(qkv)
calls = [("two-kernel", lambda: two_kernel(base, scr))] + [(f"fused[{nm}]", (lambda L: (lambda: fused(L, scr)))(L)) for nm, L in libs.items()]
t = {n: [] for n, _ in calls}
for _ in range(a.rounds):
    for n, f in calls:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            assert f() == 0
        e1.record()
        torch.cuda.synchronize()
        t[n].append(e0.elapsed_time(e1) * 1e3 / a.reps)
med = {n: sorted(v)[len(v) // 2] for n, v in t.items()}
fl = 10.0 * float((lens.double() ** 2).sum()) * 64 * H
for n, _ in calls:
    print(f"{n:14s} {med[n]:10.1f} us  {fl / med[n] / 1e6:7.1f} TF counted   x{med[n] / med['two-kernel']:.3f}")
