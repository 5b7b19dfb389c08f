"""Compile-time ablations of the two-workgroups-per-CU GEMM (gemm_bf16_v7.hip): what does each part of the K loop cost?
mask bits: 1 no LDS-DMA, 2 no barrier, 4 no W fragment reads, 8 no MFMA, 16 no epilogue, 32 no X fragment reads, 64 no vmcnt waits.
usage: python scripts/gemm_v7_ablate.py [--form plain|bwd] [--K 768] [--N 3072]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=2048)
ap.add_argument("--reps", type=int, default=8)
a = ap.parse_args()
lib = _C.dev_lib()
lib.cx_gemm_set_variant(6)
T, d, I = a.chunk * 128, 768, 3072
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(T, d, device=dev).bfloat16()
xl = torch.randn(T, I, device=dev).bfloat16()
w = (torch.randn(I, d, device=dev) * 0.05).bfloat16()
wl = (torch.randn(d, I, device=dev) * 0.05).bfloat16()
act, gate = torch.randn(T, I, device=dev).bfloat16(), torch.randn(T, I, device=dev).bfloat16()
out = torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16)
outd = torch.empty(T, d, device=dev, dtype=torch.bfloat16)
P = lambda t: t.data_ptr()
forms = {
    "bwd   N=3072 K=768 ": (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_swiglu_bwd_gate(P(x), P(w), P(act), P(gate), P(out), T, I, d, d, d, I, 2 * I, s)),
    "plain N=3072 K=768 ": (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_nt(P(x), P(w), P(out), None, T, I, d, d, d, I, 0, 1, 1.0, s)),
    "plain N=768  K=3072": (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_nt(P(xl), P(wl), P(outd), None, T, d, I, I, I, d, 0, 1, 1.0, s)),
}
masks = [(0, "full kernel"), (16, "no epilogue"), (17, "no epilogue, no DMA"), (19, "  + no barrier"), (83, "  + no vmcnt waits"),
         (87, "  + no W reads"), (119, "  + no X reads (MFMA only)"), (127, "nothing (loop skeleton)"), (111, "everything but MFMA + epilogue... (no DMA/bar/reads/MFMA off: skeleton+waits)"),
         (1, "no DMA"), (2, "no barrier"), (64, "no vmcnt waits"), (65, "no DMA, no vmcnt waits"), (81, "no epilogue, no DMA, no waits")]
lib.cx_gemm_v7_mode(1)
print(f"# T = {T}; us per launch (median of 3 x {a.reps}); v7 kernel ablations")
for name, (fl, run) in forms.items():
    print(name)
    for m, label in masks:
        lib.cx_gemm_v7_ablate(m)
        for _ in range(2):
            assert run() == 0
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / a.reps)
        us = sorted(ts)[1]
        print(f"   mask {m:3d} {label:48s} {us:9.1f} us  {fl / us / 1e6:8.1f} TF-equivalent")
lib.cx_gemm_v7_ablate(0)
lib.cx_gemm_v7_mode(0)
for name, (fl, run) in forms.items():
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    print(f"v6 {name} {us:9.1f} us  {fl / us / 1e6:8.1f} TF")
lib.cx_gemm_v7_mode(-1)
