"""Sweep the start stagger of the two-workgroups-per-CU GEMM (tile period over which workgroup starts are spread; 0 = all at
once, -1 = the launcher's estimate) on the launches it is meant for.  usage: python scripts/gemm_v7_stagger.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
lib.cx_gemm_set_variant(6)
T, d, I = 262144, 768, 3072
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *sh, std=1.0: (torch.randn(*sh, device=dev, generator=g) * std).bfloat16()
x, res = rn(T, d), rn(T, d)
w1, w2t, wo, wqkv = rn(2 * I, d, std=0.05), rn(I, d, std=0.05), rn(d, d, std=0.05), rn(3 * d, d, std=0.05)
act, gate = rn(T, I), rn(T, I, std=2.0)
big = torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16)
out_d = torch.empty(T, d, device=dev, dtype=torch.bfloat16)
out_I = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
gs = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
P = lambda t: t.data_ptr()
cases = {
    "fc2 dgrad + swiglu bwd": lambda: lib.cx_gemm_bf16_swiglu_bwd_gate(P(x), P(w2t), P(act), P(gate), P(big), T, I, d, d, d, I, 2 * I, s),
    "fc1 + swiglu gate save": lambda: lib.cx_gemm_bf16_swiglu_gate(P(x), P(w1), P(gs), P(out_I), T, I, d, d, d, I, I, s),
    "qkv fwd (plain)       ": lambda: lib.cx_gemm_bf16_nt(P(x), P(wqkv), P(big), None, T, 3 * d, d, d, d, 3 * d, 0, 1, 1.0, s),
    "out_proj + residual   ": lambda: lib.cx_gemm_bf16_nt_residual(P(x), P(wo), P(out_d), None, P(res), T, d, d, d, d, d, d, s),
    "fc2 dgrad (plain)     ": lambda: lib.cx_gemm_bf16_nt(P(x), P(w2t), P(out_I), None, T, I, d, d, d, I, 0, 1, 1.0, s),
}
periods = [0, 8000, 16000, 24000, 32000, 40000, 56000, 80000, 120000, -1]


def timed(run, reps=8):
    for _ in range(2):
        assert run() == 0
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[1]


print("# us per launch at T = 262144; columns = stagger period in shader cycles (0 = none, -1 = launcher estimate); last = v6")
print(f"{'launch':24s}" + "".join(f"{p:>9d}" for p in periods) + f"{'v6':>9s}")
for name, run in cases.items():
    row = []
    lib.cx_gemm_v7_mode(1)
    for per in periods:
        lib.cx_gemm_v7_period(per)
        row.append(timed(run))
    lib.cx_gemm_v7_mode(0)
    v6 = timed(run)
    print(f"{name:24s}" + "".join(f"{v:9.1f}" for v in row) + f"{v6:9.1f}")
lib.cx_gemm_v7_period(-1)
lib.cx_gemm_v7_mode(-1)
