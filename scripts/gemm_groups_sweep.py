"""N-group (XCD grid) sweep of the persistent NT GEMM: which split of the 8 XCDs over the tile matrix is fastest per shape?
(dev library: cx_gemm_v6_force_groups).  usage: python scripts/gemm_groups_sweep.py [--chunk 2048]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=2048)
a = ap.parse_args()
lib = _C.dev_lib()
T, d, I = a.chunk * 128, 768, 3072
s = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, N, K in (("qkv fwd", 3 * d, d), ("out fwd", d, d), ("fc1 fwd", 2 * I, d), ("fc2 fwd", d, I), ("fc1 dgrad", d, 2 * I),
                   ("fc2 dgrad", I, d), ("qkv dgrad", d, 3 * d)):
    x = torch.randn(T, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    y = torch.empty(T, N, device="cuda", dtype=torch.bfloat16)
    best = {}
    for _ in range(3):   # interleaved rounds, minimum per setting: the first launches after an allocation run at another clock
        for gn in (0, 1, 2, 4, 8):
            lib.cx_gemm_set_debug(gn << 8)   # bits 8..11: force the XCD-grid N-group count
            t = timeit(lambda: lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, T, N, K, K, K, N, 0, 1, 1.0, s))
            best[gn] = min(best.get(gn, 1e30), t)
    lib.cx_gemm_set_debug(0)
    print(f"{name:10s} N={N:5d} K={K:5d}   " + "   ".join(f"gn={g or 'auto'}: {t:7.1f} us" for g, t in best.items()))


# the fused-epilogue kernels of the step (fc1 + SwiGLU with / without the (y, gate) save, fc2-dgrad + SwiGLU backward): their
# epilogue streams go through the same L2 as the operand panels
xx = torch.randn(T, d, device="cuda").bfloat16()
w1 = (torch.randn(2 * I, d, device="cuda") * 0.05).bfloat16()
w2t = (torch.randn(I, d, device="cuda") * 0.05).bfloat16()
yg = torch.randn(T, 2 * I, device="cuda").bfloat16()
dyg = torch.empty(T, 2 * I, device="cuda", dtype=torch.bfloat16)
act = torch.empty(T, I, device="cuda", dtype=torch.bfloat16)
fused = {
    "fc1+swiglu save": lambda: lib.cx_gemm_bf16_swiglu(xx.data_ptr(), w1.data_ptr(), yg.data_ptr(), act.data_ptr(), T, I, d, d, d, 2 * I, I, s),
    "fc1+swiglu": lambda: lib.cx_gemm_bf16_swiglu(xx.data_ptr(), w1.data_ptr(), None, act.data_ptr(), T, I, d, d, d, 2 * I, I, s),
    "fc2dgrad+swiglu_bwd": lambda: lib.cx_gemm_bf16_swiglu_bwd(xx.data_ptr(), w2t.data_ptr(), yg.data_ptr(), dyg.data_ptr(), T, I, d, d, d, 2 * I, s),
}
for name, fn in fused.items():
    best = {}
    for _ in range(3):
        for gn in (0, 1, 2, 4, 8):
            lib.cx_gemm_set_debug(gn << 8)
            best[gn] = min(best.get(gn, 1e30), timeit(fn))
    lib.cx_gemm_set_debug(0)
    print(f"{name:20s}   " + "   ".join(f"gn={g or 'auto'}: {t:7.1f} us" for g, t in best.items()))
