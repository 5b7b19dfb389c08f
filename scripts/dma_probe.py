"""Measure global->LDS DMA throughput per CU for the access patterns of the GEMM operand tiles (cx_probe_dma_bw)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
buf = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
sink = torch.zeros(256, device="cuda")
s = torch.cuda.current_stream().cuda_stream
iters, nwg = 400, 256
print("pattern                         per_wave depth   GB/s/CU   B/clk@2.0GHz   TB/s chip")
for name, row_stride, span, wg_stride in [
    ("contiguous 1KiB, L2-resident", 128, 64 << 10, 64 << 10),
    ("contiguous 1KiB, HBM stream", 128, 7 << 20, 7 << 20),
    ("8x128B stride 1536, L2-res", 1536, 96 << 10, 96 << 10),
    ("8x128B stride 1536, stream", 1536, 7 << 20, 7 << 20),
    ("8x128B stride 12288, stream", 12288, 7 << 20, 7 << 20),
    ("8x128B stride 1536, shared W", 1536, 393216, 0),
]:
    for per_wave, depth in [(8, 1), (8, 2), (4, 2), (2, 4)]:
        args = (buf.data_ptr(), wg_stride, span, row_stride, per_wave, iters, depth, nwg, sink.data_ptr(), s)
        assert lib.cx_probe_dma_bw(*args) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.cx_probe_dma_bw(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        bytes_cu = iters * 8 * per_wave * 1024
        gbs = bytes_cu / ms / 1e6
        print(f"{name:32s} {per_wave:5d} {depth:5d} {gbs:9.1f} {gbs/2.0:12.1f} {gbs*nwg/1e3:10.2f}")
