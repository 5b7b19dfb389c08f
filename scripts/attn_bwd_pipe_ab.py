"""A/B of the fused S <= 128 attention backward: serial form (cx_attn_set_bwd_s128(3)) vs the form that requests the next
problem's rows ahead of the dQ store (mode 4).  Results must be bit-identical.  usage: python scripts/attn_bwd_pipe_ab.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
s = torch.cuda.current_stream().cuda_stream
H, D = 12, 64
for lens_name, lens in (("full 128 x 2048", [128] * 2048), ("ragged", [int(x) for x in torch.randint(1, 129, (2048,), generator=torch.Generator().manual_seed(1))])):
    B, T = len(lens), sum(lens)
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = (torch.randn(T, 3 * H * D, device="cuda", generator=g) * 0.5).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    inv = 1.0 / (1000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.outer(torch.arange(128, dtype=torch.float32), inv)
    cos, sin = torch.cos(fr).cuda().contiguous(), torch.sin(fr).cuda().contiguous()
    out = torch.empty(T, H * D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(H * T, device="cuda")
    dout = torch.randn(T, H * D, device="cuda", generator=g).bfloat16()
    delta = torch.empty(H * T, device="cuda")
    assert lib.cx_attn_varlen_fwd(qkv.data_ptr(), cu.data_ptr(), cos.data_ptr(), sin.data_ptr(), out.data_ptr(), lse.data_ptr(), B, H, T, 128, 0.125, s) == 0
    res, tm = {}, {3: [], 4: [], 13: []}
    for rnd in range(4):
        for mode in (3, 4, 13):
            lib.cx_attn_set_bwd_s128(mode % 10)
            lib.cx_attn_set_prio(1 if mode == 13 else 0)
            dqkv = torch.zeros_like(qkv)
            run = lambda: lib.cx_attn_varlen_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), cu.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                 delta.data_ptr(), dqkv.data_ptr(), B, H, T, 128, 0.125, s)
            assert run() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            tm[mode].append(e0.elapsed_time(e1) * 100)
            res[mode] = dqkv
    lib.cx_attn_set_bwd_s128(3)
    lib.cx_attn_set_prio(0)
    print(f"{lens_name:18s} T = {T:7d}: serial {sorted(tm[3])[1]:8.1f} us   pipelined {sorted(tm[4])[1]:8.1f} us   serial + setprio {sorted(tm[13])[1]:8.1f} us   bit-identical: {torch.equal(res[3], res[4]) and torch.equal(res[3], res[13])}")
