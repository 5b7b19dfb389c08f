"""Phase trace of the fused fc1 + SwiGLU GEMM (gemm_bf16_v6_kernel<SWIGLU_G>, dev library, DBG = 128 instantiation): s_memtime phase sums of
wave 0 of every workgroup over all of its tiles (see CX_PH in gemm_bf16_v6.hip).  usage: python scripts/gemm_swiglu_trace.py [--chunk 2048]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=2048)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
lib = _C.dev_lib()
lib.cx_gemm_set_variant(6)
T, d, I = a.chunk * 128, 768, 3072
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(T, d, device="cuda").bfloat16()
w1 = (torch.randn(2 * I, d, device="cuda") * 0.05).bfloat16()
gsave = torch.empty(T, I, device="cuda", dtype=torch.bfloat16)
act = torch.empty(T, I, device="cuda", dtype=torch.bfloat16)
trace = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
names = {2: "K loop (12 K-tiles per tile)", 3: "epilogue prologue: pass 0 arithmetic + staging", 4: "row-read issue (4 passes)",
         5: "next pass's accumulator read + SiLU (3 passes)", 6: "its staging writes (3 passes)", 7: "stores: 4 gate + 4 act rows per pass",
         8: "tile tail: fragment re-read, cursor, barrier"}
for save in (True, False):
    run = lambda: lib.cx_gemm_bf16_swiglu_gate(x.data_ptr(), w1.data_ptr(), gsave.data_ptr() if save else None, act.data_ptr(), T, I, d, d, d, I, I, s)
    for mask, what in ((0, "untraced"), (128, "traced")):
        lib.cx_gemm_v6_trace(trace.data_ptr() if mask else None)
        lib.cx_gemm_v6_ablate(mask)
        for _ in range(3):
            assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        print(f"fc1 + SwiGLU ({'gate save' if save else 'no save'}), T = {T}: {what:9s} {us:8.1f} us  {2.0 * T * 2 * I * d / us / 1e6:7.1f} TF")
    tr = trace.view(256, 16).cpu().double()
    tiles = tr[:, 1] / (d // 64)
    tot = tr[:, 0]
    print(f"  {float(tiles.mean()):.1f} tiles per workgroup, {float((tot / tiles).mean()):.0f} cycles per tile (s_memtime), phases per tile (mean over 256 workgroups):")
    for i in range(2, 9):
        v = float((tr[:, i] / tiles).mean())
        print(f"    {names[i]:52s} {v:8.0f}  {100 * v / float((tot / tiles).mean()):5.1f} %")
    print(f"    (inside the K loop: the counted DMA wait of the tile's first K-tile {float((tr[:, 9] / tiles).mean()):.0f}, of the other 11 together {float((tr[:, 10] / tiles).mean()):.0f}; an s_memtime pair alone reads ~{40}; the first K-tile as a whole {float((tr[:, 11] / tiles).mean()):.0f} cycles, the other 11 {float(((tr[:, 2] - tr[:, 11]) / tiles).mean()) / 11:.0f} each)")
# ---- fc2 dgrad + SwiGLU backward from (act, gate): gemm_bf16_v6_kernel<SWIGLU_BWD_AG>
dy = torch.randn(T, d, device="cuda").bfloat16()
w2t = (torch.randn(I, d, device="cuda") * 0.05).bfloat16()
gate = torch.randn(T, I, device="cuda").bfloat16()
act.copy_(torch.randn(T, I, device="cuda").bfloat16())
dyg = torch.empty(T, 2 * I, device="cuda", dtype=torch.bfloat16)
names_b = {2: "K loop (12 K-tiles per tile)", 3: "pass 0's 16 (act, gate) row loads issued", 4: "staging the loaded rows into LDS (waits for them)",
           5: "next pass's 16 row loads issued", 6: "accumulator read + SwiGLU backward on the staged cells", 7: "row reads + 16 stores per pass",
           8: "tile tail: fragment re-read, cursor, barrier"}
run = lambda: lib.cx_gemm_bf16_swiglu_bwd_gate(dy.data_ptr(), w2t.data_ptr(), act.data_ptr(), gate.data_ptr(), dyg.data_ptr(), T, I, d, d, d, I, 2 * I, s)
for mask, what in ((0, "untraced"), (128, "traced")):
    lib.cx_gemm_v6_trace(trace.data_ptr() if mask else None)
    lib.cx_gemm_v6_ablate(mask)
    for _ in range(3):
        assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    print(f"fc2 dgrad + SwiGLU backward (act, gate), T = {T}: {what:9s} {us:8.1f} us  {2.0 * T * I * d / us / 1e6:7.1f} TF")
tr = trace.view(256, 16).cpu().double()
tiles = tr[:, 1] / (d // 64)
tot = tr[:, 0]
print(f"  {float(tiles.mean()):.1f} tiles per workgroup, {float((tot / tiles).mean()):.0f} cycles per tile (s_memtime), phases per tile (mean over 256 workgroups):")
for i in range(2, 9):
    v = float((tr[:, i] / tiles).mean())
    print(f"    {names_b[i]:58s} {v:8.0f}  {100 * v / float((tot / tiles).mean()):5.1f} %")
lib.cx_gemm_v6_ablate(0)
lib.cx_gemm_v6_trace(None)
