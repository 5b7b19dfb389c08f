"""A/B of the plain one-wave-per-SIMD GEMM with its staged epilogue vs the deferred register stores (gemm_bf16_v6.hip, DEFER) on
the encoder's plain launches (HIP events, random bf16 data, one process, interleaved rounds).
usage: python scripts/gemm_defer_ab.py [--chunk 2048] [--rounds 5] [--reps 10]"""
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
a = ap.parse_args()
lib = _C.dev_lib()
lib.cx_gemm_set_variant(6)
lib.cx_gemm_v7_mode(0)
T, d, I = a.chunk * 128, 768, 3072
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *sh, std=1.0: (torch.randn(*sh, device=dev, generator=g) * std).bfloat16()
x, x3, xI, x2I = rn(T, d), rn(T, 3 * d), rn(T, I), rn(T, 2 * I)
w1 = rn(2 * I, d, std=0.05)
w1t = rn(d, 2 * I, std=0.05)
w2t = rn(I, d, std=0.05)
w2 = rn(d, I, std=0.05)
wo, wqkv, wqkv_t = rn(d, d, std=0.05), rn(3 * d, d, std=0.05), rn(d, 3 * d, std=0.05)
out_d = torch.empty(T, d, device=dev, dtype=torch.bfloat16)
out_3d = torch.empty(T, 3 * d, device=dev, dtype=torch.bfloat16)
out_I = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
out_2I = torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16)
P = lambda t: t.data_ptr()
nt = lambda X, W, O, N, K: (2.0 * T * N * K, lambda: lib.cx_gemm_bf16_nt(P(X), P(W), P(O), None, T, N, K, K, K, N, 0, 1, 1.0, s))
cases = {
    "qkv fwd                 N=2304 K=768": nt(x, wqkv, out_3d, 3 * d, d),
    "out_proj (plain)        N= 768 K=768": nt(x, wo, out_d, d, d),
    "fc2 dgrad (plain)       N=3072 K=768": nt(x, w2t, out_I, I, d),
    "fc1 (plain)             N=6144 K=768": nt(x, w1, out_2I, 2 * I, d),
    "qkv dgrad (plain)       N= 768 K=2304": nt(x3, wqkv_t, out_d, d, 3 * d),
    "fc2 fwd (plain)         N= 768 K=3072": nt(xI, w2, out_d, d, I),
    "fc1 dgrad (plain)       N= 768 K=6144": nt(x2I, w1t, out_d, d, 2 * I),
}
print(f"# T = {T} token rows per launch; times are the median of {a.rounds} interleaved rounds of {a.reps} launches (us)")
print(f"{'launch':40s} {'staged us':>10s} {'TF':>8s} {'deferred us':>12s} {'TF':>8s} {'def/staged':>10s} {'bit-identical':>14s}")
for name, (fl, run) in cases.items():
    if a.only and a.only not in name:
        continue
    t = {0: [], 1: []}
    outs = {}
    for mode in (0, 1):
        lib.cx_gemm_v6_defer(mode)
        for _ in range(2):
            assert run() == 0, name
        torch.cuda.synchronize()
        o = out_3d if "qkv fwd" in name else out_I if "fc2 dgrad" in name else out_2I if "fc1 (plain)" in name else out_d
        outs[mode] = o.clone()
    for _ in range(a.rounds):
        for mode in (0, 1):
            lib.cx_gemm_v6_defer(mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    m0, m1 = sorted(t[0])[len(t[0]) // 2], sorted(t[1])[len(t[1]) // 2]
    print(f"{name:40s} {m0:10.1f} {fl / m0 / 1e6:8.1f} {m1:12.1f} {fl / m1 / 1e6:8.1f} {m1 / m0:10.3f} {str(torch.equal(outs[0], outs[1])):>14s}")
lib.cx_gemm_v6_defer(-1)
lib.cx_gemm_v7_mode(-1)
