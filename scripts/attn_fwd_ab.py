"""A/B of the S <= 128 attention forward kernels (cx_attn_set_fwd_s128 modes) at the metric's shape, plus a ragged
batch for the masked path.  usage: python scripts/attn_fwd_ab.py [modes...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

modes = [int(m) for m in sys.argv[1:]] or [0, 1, 2]
lib = _C.dev_lib()
s = torch.cuda.current_stream().cuda_stream
H, D, S = 12, 64, 128
inv = 1.0 / (1000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
cos, sin = torch.cos(fr).cuda().contiguous(), torch.sin(fr).cuda().contiguous()
for ragged in (False, True):
    B = 1024
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(1, S + 1, (B,), generator=g) if ragged else torch.full((B,), S)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    T = int(cu[-1])
    cu = cu.cuda()
    qkv = (torch.randn(T, 3 * H * D, device="cuda") * 0.5).to(torch.bfloat16)
    ref = None
    for rep in range(2):
        for mode in modes:
            lib.cx_attn_set_fwd_s128(mode)
            out = torch.zeros(T, H * D, device="cuda", dtype=torch.bfloat16)
            lse = torch.zeros(H * T, device="cuda")
            fwd = lambda: lib.cx_attn_varlen_fwd(qkv.data_ptr(), cu.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                 out.data_ptr(), lse.data_ptr(), B, H, T, S, 0.125, s)
            assert fwd() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fwd()
            e1.record()
            torch.cuda.synchronize()
            if ref is None:
                ref = (out.float().clone(), lse.clone())
            d_o = float((out.float() - ref[0]).abs().max())
            d_l = float((lse - ref[1]).abs().max())
            print(f"ragged={ragged} mode {mode}: {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us   max|dO| vs mode {modes[0]} = {d_o:.2e}"
                  f"  max|dlse| = {d_l:.2e}")
lib.cx_attn_set_fwd_s128(0)
