"""Phase timers of the fused 128 < S <= 256 attention backward (attn_bwd_s256_kernel built with -DCX_ATTN_TRACE=1): thread 0 of every
workgroup stamps s_memtime at the phase boundaries of its first 16 problems (vmcnt / lgkmcnt drained at every stamp).
usage: python scripts/build_variant.py attntrace attention.hip -DCX_ATTN_TRACE=1
       CONTRASTORS_HIP_DEV_LIB=contrastors_amd/lib/variants/libcontrastors_hip_dev_attntrace.so python scripts/attn_trace_s256.py [S]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
lib.cx_attn_set_bwd_long(0)
lib.cx_attn_set_bwd_s256(1)
s = torch.cuda.current_stream().cuda_stream
S = int(sys.argv[1]) if len(sys.argv) > 1 else 197
H, D = 12, 64
B = 262144 // S
T = B * S
qkv = (torch.randn(T, 3 * H * D, device="cuda") * 0.5).to(torch.bfloat16)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
out = torch.empty(T, H * D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(H * T, device="cuda")
dout = torch.randn_like(out)
dqkv = torch.empty_like(qkv)
delta = torch.zeros(max(H * T, 256 * 16 * 16 * 2), device="cuda")
assert lib.cx_attn_varlen_fwd(qkv.data_ptr(), cu.data_ptr(), None, None, out.data_ptr(), lse.data_ptr(), B, H, T, S, 0.125, s) == 0
bwd = lambda: lib.cx_attn_varlen_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), cu.data_ptr(), None, None,   # noqa: E731
                                     delta.data_ptr(), dqkv.data_ptr(), B, H, T, S, 0.125, s)
for _ in range(3):
    assert bwd() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
bwd()
e1.record()
torch.cuda.synchronize()
print(f"S = {S}: bwd with stamps {e0.elapsed_time(e1) * 1e3:.1f} us for T = {T} ({B * H} problems on 256 workgroups)")
tr = delta.view(torch.int64)[: 256 * 16 * 16].view(256, 16, 16).cpu().double()
names = ["K, V, K^T staged (loads landed)", "half 0 staged (Q^T | dO^T, delta)", "barrier A, key fragments, barrier B", "main loop, half 0",
         "barrier C", "dQ products + exchange (+ next half's rows landed)", "barrier D, dQ add + store, barrier E",
         "half 1 staged", "barrier A", "main loop, half 1", "barrier C", "dQ products + exchange", "barrier D, dQ add + store, barrier E",
         "dK, dV stored", "barrier F"]
d = tr[:, 2:14, 1:16] - tr[:, 2:14, 0:15]           # problems 2..13 of every workgroup
tot = (tr[:, 2:14, 15] - tr[:, 2:14, 0]).mean().item()
gap = (tr[:, 3:14, 0] - tr[:, 2:13, 15]).mean().item()
print(f"cycles per problem (thread 0, mean over 256 workgroups x 12 problems): total {tot:.0f}, between problems {gap:.0f}")
for i, n in enumerate(names):
    m = d[:, :, i]
    print(f"  {n:52s} {m.mean().item():8.0f}  (p10 {m.flatten().quantile(0.1).item():7.0f}  p90 {m.flatten().quantile(0.9).item():7.0f})  {100 * m.mean().item() / tot:5.1f} %")
