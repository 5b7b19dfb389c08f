"""Phase breakdown of the persistent 256x256x64 GEMM (cx_gemm_set_trace): per-wave cycles spent waiting for operands
(vmcnt + barrier), in the LDS-read/MFMA phase and in the epilogue.  usage: python scripts/gemm_trace.py [--chunk 512]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=512)
ap.add_argument("--dbg", type=int, default=0)
ap.add_argument("--zeros", type=int, default=0)
ap.add_argument("--shape", type=str, action="append", default=[], help="M,N,K (repeatable)")
a = ap.parse_args()
lib = _C.dev_lib()
lib.cx_gemm_set_debug(a.dbg)
T = a.chunk * 128
shapes = {"qkv_fwd": (T, 2304, 768), "out_fwd": (T, 768, 768), "fc1_fwd": (T, 6144, 768), "fc2_fwd": (T, 768, 3072),
          "fc1_dgrad": (T, 768, 6144), "fc2_dgrad": (T, 3072, 768), "qkv_dgrad": (T, 768, 2304)}
if a.shape:
    shapes = {f"custom{i}": tuple(int(v) for v in sh.split(",")) for i, sh in enumerate(a.shape)}
s = torch.cuda.current_stream().cuda_stream
trace = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
print("shape        us      TF   | per wave, cycles/iteration: wait  compute  epilogue(amortised)  | MFMA-bound 2048")
for name, (M, N, K) in shapes.items():
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    if a.zeros:
        x.zero_(); w.zero_()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    run = lambda: lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, K, K, K, N, 0, 1, 1.0, s)
    for _ in range(2):
        assert run() == 0
    lib.cx_gemm_set_trace(trace.data_ptr())
    trace.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert run() == 0
    e1.record()
    torch.cuda.synchronize()
    lib.cx_gemm_set_trace(None)
    us = e0.elapsed_time(e1) * 1e3
    t = trace.view(256, 8, 8).double()
    t = t[t[..., 3].sum(1) > 0]
    it = t[..., 3].clamp(min=1)
    wait, comp, epi = (t[..., 0] / it).mean().item(), (t[..., 1] / it).mean().item(), (t[..., 2] / it).mean().item()
    tot = (t[..., 0] + t[..., 1] + t[..., 2]).max().item()
    print(f"{name:10s} {us:7.1f} {2.0*M*N*K/us/1e6:7.1f} | {wait:7.0f} {comp:7.0f} {epi:7.0f}   iters/wave {it.mean().item():.0f}  epi/tile {epi * K / 64:.0f} (dma-wait {(t[..., 4] / it).mean().item() * K / 64:.0f})"
          f"  max wave total {tot/1e3:.0f} kcyc -> {tot/us/1e3:.2f} GHz")
