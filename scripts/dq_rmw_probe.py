"""VERDICT r4 item 4, measured instead of priced: the dQ accumulation traffic of a single-owner fused long-sequence attention
backward BY ITSELF (cx_probe_rmw, dev library).  cfg 3's per-GPU shape: 288 sequences x 2048 tokens x 12 heads = 3456 problems;
one workgroup owns a problem, dK / dV of a 128-key block live in registers while the query blocks run in the inner loop, and every
(key block, query block) pair adds a 128 x 64 fp32 partial into the problem's dQ scratch: 16 key blocks = 16 load + add + store
sweeps over 2048 x 64 fp32 = 512 KB that only this workgroup touches.  Compared with what the two-kernel backward takes for the
whole layer call (scripts/attn_microbench.py, S = 2048)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
s = torch.cuda.current_stream().cuda_stream
print("S      problems  key-blocks  scratch GB   RMW-only ms   traffic GB   TB/s   (256 workgroups, one problem at a time each)")
for S, seqs in ((2048, 288), (512, 1152), (197, 4096)):
    H = 12
    n = seqs * H
    floats = S * 64
    sweeps = (S + 127) // 128
    buf = torch.zeros(n * floats, device="cuda")
    for nwg in (256, 512):
        assert lib.cx_probe_rmw(buf.data_ptr(), floats, sweeps, n, nwg, s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            lib.cx_probe_rmw(buf.data_ptr(), floats, sweeps, n, nwg, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        gb = 2.0 * n * floats * 4 * sweeps / 1e9
        print(f"{S:5d} {n:9d} {sweeps:10d} {n * floats * 4 / 1e9:11.2f} {ms:13.2f} {gb:12.1f} {gb / ms:6.2f}   nwg={nwg}")
    del buf
