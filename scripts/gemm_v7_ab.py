"""A/B of the two NT GEMM structures on the encoder's launches (HIP events, random bf16 data, one process, interleaved rounds):
v6 = one wave per SIMD, 256x256 tiles (gemm_bf16_v6.hip); v7 = two workgroups per CU, 256x128 tiles (gemm_bf16_v7.hip).
usage: python scripts/gemm_v7_ab.py [--chunk 2048] [--rounds 5] [--reps 10]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=2048)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--only", type=str, default="")
ap.add_argument("--flags", type=int, default=0, help="cx_gemm_v7_flags (bit 0: K loop at s_setprio 1)")
a = ap.parse_args()
lib = _C.dev_lib()
lib.cx_gemm_set_variant(6)
lib.cx_gemm_v7_flags(a.flags)
T, d, I = a.chunk * 128, 768, 3072
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *sh, std=1.0: (torch.randn(*sh, device=dev, generator=g) * std).bfloat16()
x, res = rn(T, d), rn(T, d)
x3 = rn(T, 3 * d)
w1 = rn(2 * I, d, std=0.05)
w2t = rn(I, d, std=0.05)
w2 = rn(d, I, std=0.05)
wo, wqkv, wqkv_t = rn(d, d, std=0.05), rn(3 * d, d, std=0.05), rn(d, 3 * d, std=0.05)
act, gate = rn(T, I), rn(T, I, std=2.0)
dyg = torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16)
out_d = torch.empty(T, d, device=dev, dtype=torch.bfloat16)
out_3d = torch.empty(T, 3 * d, device=dev, dtype=torch.bfloat16)
out_I = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
gs = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
P = lambda t: t.data_ptr()
cases = {
    "fc2 dgrad + swiglu bwd (act, gate)  N=3072 K=768": (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_swiglu_bwd_gate(P(x), P(w2t), P(act), P(gate), P(dyg), T, I, d, d, d, I, 2 * I, s)),
    "fc1 + swiglu, gate save             N=6144 K=768": (2.0 * T * 2 * I * d, lambda: lib.cx_gemm_bf16_swiglu_gate(P(x), P(w1), P(gs), P(out_I), T, I, d, d, d, I, I, s)),
    "fc1 + swiglu, no save               N=6144 K=768": (2.0 * T * 2 * I * d, lambda: lib.cx_gemm_bf16_swiglu_gate(P(x), P(w1), None, P(out_I), T, I, d, d, d, I, I, s)),
    "qkv fwd                             N=2304 K=768": (2.0 * T * 3 * d * d, lambda: lib.cx_gemm_bf16_nt(P(x), P(wqkv), P(out_3d), None, T, 3 * d, d, d, d, 3 * d, 0, 1, 1.0, s)),
    "out_proj fwd + residual             N= 768 K=768": (2.0 * T * d * d, lambda: lib.cx_gemm_bf16_nt_residual(P(x), P(wo), P(out_d), None, P(res), T, d, d, d, d, d, d, s)),
    "out_proj dgrad (plain)              N= 768 K=768": (2.0 * T * d * d, lambda: lib.cx_gemm_bf16_nt(P(x), P(wo), P(out_d), None, T, d, d, d, d, d, 0, 1, 1.0, s)),
    "fc2 dgrad (plain)                   N=3072 K=768": (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_nt(P(x), P(w2t), P(out_I), None, T, I, d, d, d, I, 0, 1, 1.0, s)),
    "qkv dgrad + residual                N= 768 K=2304": (2.0 * T * 3 * d * d, lambda: lib.cx_gemm_bf16_nt_residual(P(x3), P(wqkv_t), P(out_d), None, P(res), T, d, 3 * d, 3 * d, 3 * d, d, d, s)),
    "fc2 fwd + residual                  N= 768 K=3072": (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_nt_residual(P(act), P(w2), P(out_d), None, P(res), T, d, I, I, I, d, d, s)),
}
print(f"# T = {T} token rows per launch; times are the median of {a.rounds} interleaved rounds of {a.reps} launches (us)")
print(f"{'launch':52s} {'v6 us':>9s} {'v6 TF':>8s} {'v7 us':>9s} {'v7 TF':>8s} {'v7/v6':>7s}")
for name, (fl, run) in cases.items():
    if a.only and a.only not in name:
        continue
    t = {0: [], 1: []}
    for mode in (0, 1):
        lib.cx_gemm_v7_mode(mode)
        for _ in range(2):
            assert run() == 0, name
    for _ in range(a.rounds):
        for mode in (0, 1):
            lib.cx_gemm_v7_mode(mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    m6, m7 = sorted(t[0])[len(t[0]) // 2], sorted(t[1])[len(t[1]) // 2]
    print(f"{name:52s} {m6:9.1f} {fl / m6 / 1e6:8.1f} {m7:9.1f} {fl / m7 / 1e6:8.1f} {m7 / m6:7.3f}")
lib.cx_gemm_v7_mode(-1)
