"""Calibration only (never on the product path): what the vendor BLAS behind torch.matmul reaches on the step's GEMM
shapes on this box, next to the shipped kernel.  usage: python scripts/blas_calibration.py [--chunk 2048]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=2048)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
T, d, I = a.chunk * 128, 768, 3072
lib = _C.lib()
dev = "cuda"


def timeit(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print(f"T = {T} token rows; bf16 x bf16 -> bf16, fp32 accumulate")
for name, N, K in (("qkv fwd", 3 * d, d), ("out fwd", d, d), ("fc1 fwd (no SwiGLU)", 2 * I, d), ("fc2 fwd", d, I),
                   ("fc1 dgrad", d, 2 * I), ("fc2 dgrad", I, d), ("qkv dgrad", d, 3 * d)):
    x = torch.randn(T, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    y = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    t_blas = timeit(lambda: torch.matmul(x, w.t(), out=y), a.reps)
    t_cx = timeit(lambda: lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, T, N, K, K, K, N, 0, 1, 1.0, s),
                  a.reps)
    fl = 2.0 * T * N * K
    print(f"{name:22s} N={N:5d} K={K:5d}   vendor BLAS {t_blas:8.1f} us {fl / t_blas / 1e6:7.1f} TF   |   v6 {t_cx:8.1f} us "
          f"{fl / t_cx / 1e6:7.1f} TF")
# wgrad form: dW (N, K) = dY^T (N, T) @ X (T, K)
for name, N, K in (("qkv wgrad", 3 * d, d), ("fc1 wgrad", 2 * I, d), ("fc2 wgrad", d, I)):
    dy = torch.randn(T, N, device=dev).bfloat16()
    x = torch.randn(T, K, device=dev).bfloat16()
    gw = torch.zeros(N, K, device=dev)
    ws = torch.empty(16 * N * K, device=dev)
    out = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    t_blas = timeit(lambda: torch.matmul(dy.t(), x, out=out), a.reps)
    t_cx = timeit(lambda: lib.cx_gemm_bf16_tn_accum(dy.data_ptr(), x.data_ptr(), gw.data_ptr(), ws.data_ptr(), ws.numel(), T, N, K,
                                                    N, K, s), a.reps)
    fl = 2.0 * T * N * K
    print(f"{name:22s} N={N:5d} K={K:5d}   vendor BLAS {t_blas:8.1f} us {fl / t_blas / 1e6:7.1f} TF   |   v6tn {t_cx:8.1f} us "
          f"{fl / t_cx / 1e6:7.1f} TF")
