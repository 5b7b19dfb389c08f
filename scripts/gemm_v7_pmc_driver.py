"""Workload for the rocprofv3 counter passes of the v6 / v7 A/B (scripts/gpu_r4_v7_pmc.sh): the fc2-dgrad + SwiGLU-backward
launch and the plain fc2-dgrad launch of the metric's shape, 4 times each on each kernel structure."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
lib.cx_gemm_set_variant(6)
lib.cx_gemm_v7_flags(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
T, d, I = 262144, 768, 3072
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *sh, std=1.0: (torch.randn(*sh, device=dev, generator=g) * std).bfloat16()
x, w2t, act, gate = rn(T, d), rn(I, d, std=0.05), rn(T, I), rn(T, I, std=2.0)
dyg = torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16)
out_I = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
P = lambda t: t.data_ptr()
for mode in (0, 1):
    lib.cx_gemm_v7_mode(mode)
    for _ in range(4):
        assert lib.cx_gemm_bf16_swiglu_bwd_gate(P(x), P(w2t), P(act), P(gate), P(dyg), T, I, d, d, d, I, 2 * I, s) == 0
        assert lib.cx_gemm_bf16_nt(P(x), P(w2t), P(out_I), None, T, I, d, d, d, I, 0, 1, 1.0, s) == 0
torch.cuda.synchronize()
lib.cx_gemm_v7_mode(-1)
