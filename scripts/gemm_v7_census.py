"""Residency census of the two-workgroups-per-CU GEMM (gemm_bf16_v7.hip): does the hardware really keep two of its 80-KiB,
256-register workgroups on every CU?  Every workgroup records s_memtime at entry and exit plus HW_REG_HW_ID / HW_REG_XCC_ID;
two workgroups are co-resident on a CU when they report the same (XCC, SE, SH, CU) id and overlapping time spans.
usage: python scripts/gemm_v7_census.py"""
import sys
from collections import Counter, defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
lib.cx_gemm_set_variant(6)
print("occupancy API: workgroups per CU =", lib.cx_gemm_v7_occupancy())
T, d, I = 262144, 768, 3072
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(T, d, device=dev).bfloat16()
w = (torch.randn(I, d, device=dev) * 0.05).bfloat16()
act, gate = torch.randn(T, I, device=dev).bfloat16(), torch.randn(T, I, device=dev).bfloat16()
out = torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16)
tr = torch.zeros(512, 4, dtype=torch.int64, device=dev)
lib.cx_gemm_v7_mode(1)
lib.cx_gemm_v7_trace(tr.data_ptr())
for _ in range(2):
    assert lib.cx_gemm_bf16_swiglu_bwd_gate(x.data_ptr(), w.data_ptr(), act.data_ptr(), gate.data_ptr(), out.data_ptr(), T, I, d, d, d, I, 2 * I, s) == 0
torch.cuda.synchronize()
lib.cx_gemm_v7_trace(None)
lib.cx_gemm_v7_mode(-1)
t = tr.cpu().numpy()
cu = defaultdict(list)
for b in range(512):
    hw, xcc = int(t[b, 2]), int(t[b, 3]) & 0xF
    key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF)   # (XCC, SE, SH, CU): HW_ID layout of gfx9
    cu[key].append((int(t[b, 0]), int(t[b, 1]), b))
per_cu = Counter(len(v) for v in cu.values())
print(f"{len(cu)} distinct (XCC, SE, SH, CU) ids; workgroups per id: {dict(per_cu)}")
ov = 0
for v in cu.values():
    v.sort()
    for i in range(len(v) - 1):
        if v[i + 1][0] < v[i][1]:
            ov += 1
print(f"pairs of workgroups on one CU with overlapping lifetimes: {ov}")
span = t[:, 1] - t[:, 0]
print(f"workgroup lifetime (s_memtime ticks, 100 MHz): min {span.min()} median {int(sorted(span)[256])} max {span.max()}; launch first start -> last end {t[:,1].max() - t[:,0].min()}")
print("xcc of blocks 0..15:", [int(v) & 0xF for v in t[:16, 3]])
