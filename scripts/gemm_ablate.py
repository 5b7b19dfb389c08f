"""Where do the cycles of the one-wave-per-SIMD NT GEMM go?  Ablation builds of gemm_bf16_v6_kernel<NONE> with an
in-kernel s_memtime span per workgroup (cycles per K-tile are clock-independent; wall time is printed beside them).
usage: python scripts/gemm_ablate.py [--chunk 1024]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=1024)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
lib = _C.dev_lib()
lib.cx_gemm_set_variant(6)
T = a.chunk * 128
shapes = {"fc1_dgrad K=6144": (T, 768, 6144), "fc1_fwd K=768": (T, 6144, 768), "out_fwd K=768": (T, 768, 768)}
masks = [(128, "full kernel (trace only)"), (128, "full kernel again"), (64, "epilogue without global stores"), (16, "no epilogue"), (1, "no DMA"), (2, "no barrier"), (4, "no fragment reads"),
         (32, "no DMA wait"), (33, "no DMA, no wait"), (35, "no DMA/wait/barrier"), (39, "MFMA only (+epilogue)"),
         (55, "MFMA only, no epilogue"), (59, "reads only: no DMA/wait/barrier/MFMA/epilogue")]
s = torch.cuda.current_stream().cuda_stream
trace = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
lib.cx_gemm_v6_trace(trace.data_ptr())
for name, (M, N, K) in shapes.items():
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    print(f"== {name}  M={M} N={N}")
    for mask, what in masks:
        lib.cx_gemm_v6_ablate(mask)
        run = lambda: lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, K, K, K, N, 0, 1, 1.0, s)
        for _ in range(2):
            assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        tr = trace.view(256, 16).cpu()
        cyc, kt = tr[:, 0].double(), tr[:, 1].double()
        per_kt = float((cyc / kt.clamp(min=1)).mean())
        nk = K // 64
        tiles_per_wg = float(kt.mean()) / nk
        print(f"  {what:48s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF   {per_kt:7.0f} cycles/K-tile "
              f"({per_kt * nk:8.0f} per output tile, {tiles_per_wg:.1f} tiles/WG)  clock {float(cyc.mean()) / us / 1e3:.2f} GHz")
lib.cx_gemm_v6_ablate(0)
lib.cx_gemm_v6_trace(None)
