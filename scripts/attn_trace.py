"""Phase timers of the fused S <= 128 attention backward (attn_bwd_fused2_s128_kernel built with -DCX_ATTN_TRACE=1):
wave 0 of every workgroup stamps s_memtime at the phase boundaries of its first 16 problems.
usage: python scripts/build_variant.py attntrace attention.hip -DCX_ATTN_TRACE=1
       CONTRASTORS_HIP_DEV_LIB=contrastors_amd/lib/variants/libcontrastors_hip_dev_attntrace.so python scripts/attn_trace.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
s = torch.cuda.current_stream().cuda_stream
H, D, S, B = 12, 64, 128, 2048
T = B * S
qkv = (torch.randn(T, 3 * H * D, device="cuda") * 0.5).to(torch.bfloat16)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
inv = 1.0 / (1000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
cos, sin = torch.cos(fr).cuda().contiguous(), torch.sin(fr).cuda().contiguous()
out = torch.empty(T, H * D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(H * T, device="cuda")
dout = torch.randn_like(out)
dqkv = torch.empty_like(qkv)
delta = torch.zeros(H * T, device="cuda")
assert lib.cx_attn_varlen_fwd(qkv.data_ptr(), cu.data_ptr(), cos.data_ptr(), sin.data_ptr(), out.data_ptr(), lse.data_ptr(),
                              B, H, T, S, 0.125, s) == 0
bwd = lambda: lib.cx_attn_varlen_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), cu.data_ptr(),
                                     cos.data_ptr(), sin.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, H, T, S, 0.125, s)
for _ in range(3):
    assert bwd() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
bwd()
e1.record()
torch.cuda.synchronize()
print(f"bwd with stamps: {e0.elapsed_time(e1) * 1e3:.1f} us for T = {T}")
tr = delta.view(torch.int64)[: 512 * 16 * 16].view(512, 16, 16).cpu().double()
names = ["loads landed", "staged (K,V rows, delta, Qt, dOt)", "barrier", "key fragments + barrier", "main loop (4 query blocks)",
         "barrier", "dK, dV stored", "Kt staged", "barrier", "dQ products", "dQ stored", "barrier"]
d = tr[:, 2:14, 1:13] - tr[:, 2:14, 0:12]           # problems 2..13 of every workgroup
tot = (tr[:, 2:14, 12] - tr[:, 2:14, 0]).mean().item()
gap = (tr[:, 3:14, 0] - tr[:, 2:13, 12]).mean().item()
print(f"cycles per problem (wave 0, mean over 512 workgroups x 12 problems): total {tot:.0f}, between problems {gap:.0f}")
for i, n in enumerate(names):
    m = d[:, :, i]
    print(f"  {n:38s} {m.mean().item():8.0f}  (p10 {m.flatten().quantile(0.1).item():7.0f}  p90 {m.flatten().quantile(0.9).item():7.0f})  {100 * m.mean().item() / tot:5.1f} %")
