"""MFMA issue rate and sustained clock under load (cx_probe_mfma_rate): the ceiling any GEMM main loop can reach."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
s = torch.cuda.current_stream().cuda_stream
cyc = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, device="cuda")
iters = 20000
print("form      data     waves/CU  nCU   ms      cycles/MFMA/SIMD   clock GHz   TFLOP/s")
for form, fn, per_it in (("32x32x16", lib.cx_probe_mfma_rate, 8), ("16x16x32", lib.cx_probe_mfma_rate16, 16)):
    for data in ("zeros", "randn"):
        seed = (torch.zeros if data == "zeros" else torch.randn)(2048 * 8, device="cuda").to(torch.bfloat16)
        for waves, nwg in [(4, 256), (8, 256), (8, 32), (4, 32)]:
            args = (seed.data_ptr(), waves, iters, nwg, cyc.data_ptr(), sink.data_ptr(), s)
            assert fn(*args) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert fn(*args) == 0
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            c = cyc.view(256, 8)[:nwg, :waves].double().mean().item()
            per_simd = iters * per_it * (waves / 4)
            flops = nwg * waves * iters * 8 * 32 * 32 * 16 * 2
            print(f"{form:9s} {data:8s} {waves:5d} {nwg:6d} {ms:7.2f} {c / per_simd:14.1f} {c / ms / 1e6:14.2f} {flops / ms / 1e9:10.1f}")
