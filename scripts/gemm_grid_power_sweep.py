"""is HBM / L2-miss TRAFFIC paid for in CLOCK on this power-bound part?  For the fused fc1 + SwiGLU launch
(gemm_bf16_v6_kernel<5,0>: X re-fetched by every XCD of a grid row) and two plain launches (<0,0>), per XCD grid gn (N-groups of the
8-XCD grid over the tile matrix; gn = 8: an X panel is fetched by all 8 XCDs, gn = 1: by one):
   time per launch, shader clock and socket power sampled through librocm_smi64 DURING ~1.5 s of back-to-back launches.
With --pmc CASE GN the script only issues 4 launches of that (case, grid) so that
   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... python scripts/gemm_grid_power_sweep.py --pmc fc1_swiglu_save 4
attributes the counter to that grid (round 5's driver script, since removed, looped over the grids and joined both tables).
usage: python scripts/gemm_grid_power_sweep.py [--chunk 2048] [--seconds 1.5]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402
from scripts.box_calibration import SmiSampler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=2048)
ap.add_argument("--seconds", type=float, default=1.5)
ap.add_argument("--pmc", nargs=2, default=None, metavar=("CASE", "GN"))
a = ap.parse_args()
lib = _C.dev_lib()
T, d, I = a.chunk * 128, 768, 3072
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *sh, std=1.0: (torch.randn(*sh, device="cuda", generator=g) * std).bfloat16()   # noqa: E731
P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
x, w1, wqkv, w2 = rn(T, d), rn(2 * I, d, std=0.05), rn(3 * d, d, std=0.05), rn(d, I, std=0.05)
act = rn(T, I)
gs = torch.empty(T, I, device="cuda", dtype=torch.bfloat16)
out_I = torch.empty(T, I, device="cuda", dtype=torch.bfloat16)
out_3d = torch.empty(T, 3 * d, device="cuda", dtype=torch.bfloat16)
out_d = torch.empty(T, d, device="cuda", dtype=torch.bfloat16)
cases = {   # name: (flop, algorithmic bytes, call)
    "fc1_swiglu_save": (2.0 * T * 2 * I * d, 2.0 * (T * d + 2 * I * d + 2 * T * I), lambda: lib.cx_gemm_bf16_swiglu_gate(P(x), P(w1), P(gs), P(out_I), T, I, d, d, d, I, I, s)),
    "qkv_fwd": (2.0 * T * 3 * d * d, 2.0 * (T * d + 3 * d * d + 3 * T * d), lambda: lib.cx_gemm_bf16_nt(P(x), P(wqkv), P(out_3d), None, T, 3 * d, d, d, d, 3 * d, 0, 1, 1.0, s)),
    "fc2_fwd": (2.0 * T * I * d, 2.0 * (T * I + I * d + T * d), lambda: lib.cx_gemm_bf16_nt(P(act), P(w2), P(out_d), None, T, d, I, I, I, d, 0, 1, 1.0, s)),
}
if a.pmc:
    name, gn = a.pmc[0], int(a.pmc[1])
    lib.cx_gemm_set_debug(gn << 8)
    for _ in range(4):
        assert cases[name][2]() == 0
    torch.cuda.synchronize()
    sys.exit(0)

print(f"# T = {T} rows; ~{a.seconds} s of back-to-back launches per (case, grid); clock / power: librocm_smi64 at 20 Hz during that window")
print(f"{'case':18s} {'gn':>4s} {'us/launch':>10s} {'TFLOP/s':>9s} {'sclk MHz':>9s} {'power W':>8s} {'alg GB':>7s}")
for name, (fl, by, call) in cases.items():
    for gn in (0, 1, 2, 4, 8):
        lib.cx_gemm_set_debug(gn << 8)   # bits 8..11: force the XCD-grid N-group count (0 = the shipped heuristic)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        smp = SmiSampler(torch.cuda.current_device(), hz=20.0).start()
        ts = []
        t_end = time.perf_counter() + a.seconds
        while time.perf_counter() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 8)
        r = smp.stop()
        ts.sort()
        us = ts[len(ts) // 2]
        f = lambda v, fmt: (fmt % v) if v is not None else "n/a"   # noqa: E731
        print(f"{name:18s} {gn or 'auto':>4} {us:10.1f} {fl / us / 1e6:9.1f} {f(r['mean_sclk_mhz'], '%9.0f'):>9s} {f(r['mean_power_w'], '%8.0f'):>8s} {by / 1e9:7.2f}", flush=True)
lib.cx_gemm_set_debug(0)
