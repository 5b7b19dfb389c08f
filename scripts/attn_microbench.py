"""Time cx_attn_varlen_fwd / bwd (head_dim 64, non-causal, rotary on) at a few sequence lengths.
usage: python scripts/attn_microbench.py [--tokens 131072]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=131072)
ap.add_argument("--heads", type=int, default=12)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--bwd-mode", type=int, default=None, help="cx_attn_set_bwd_s128 (dev library): 4 fused3 (default), 3 fused2, ...")
ap.add_argument("--seqs", type=str, default="128,197,512,2048,8192")
ap.add_argument("--max-seqlen-pad", type=int, default=0, help="pass max_seqlen = S + this to forward and backward (A/B: S + pad > 256 selects the streaming kernels for S = 197)")
ap.add_argument("--warm-seconds", type=float, default=0.6, help="back-to-back launches before the first timed shape: the package at its sustained clock, not its boost clock (VERDICT r5 item 5a)")
ap.add_argument("--pdrop", type=float, default=0.0, help="> 0: the attention-dropout entry points (cx_attn_varlen_dropout_fwd / _bwd)")
ap.add_argument("--fwd-mode", type=int, default=None, help="cx_attn_set_fwd_s128 (dev library); with --pdrop: 0 = general kernel")
ap.add_argument("--power", type=float, default=0.0, help="> 0: after timing a kernel, this many seconds of back-to-back launches with socket power / shader clock sampled (librocm_smi64)")
ap.add_argument("--rotary", type=int, default=1, help="0: no rotation tables (image towers; what the engine passes for pre-rotated long sequences)")
ap.add_argument("--fwd-long", type=int, default=None, help="cx_attn_set_fwd_long (dev library): 0 = round 1's streaming forward for S > 256")
ap.add_argument("--bwd-long", type=int, default=None, help="cx_attn_set_bwd_long (dev library): 0 = round 1's delta + dQ + dK/dV kernels for S > 128")
ap.add_argument("--bwd-s256", type=int, default=None, help="cx_attn_set_bwd_s256 (dev library): 1 = the fused persistent backward for 128 < S <= 256 (with --bwd-long 0)")
a = ap.parse_args()
lib = _C.dev_lib()
if a.bwd_long is not None:
    lib.cx_attn_set_bwd_long(a.bwd_long)
if a.bwd_s256 is not None:
    lib.cx_attn_set_bwd_s256(a.bwd_s256)
if a.fwd_long is not None:
    lib.cx_attn_set_fwd_long(a.fwd_long)
if a.bwd_mode is not None:
    lib.cx_attn_set_bwd_s128(a.bwd_mode)
if a.fwd_mode is not None:
    lib.cx_attn_set_fwd_s128(a.fwd_mode)
s = torch.cuda.current_stream().cuda_stream
H, D = a.heads, 64
print("S      B     fwd us   fwd TF    bwd us   bwd TF   (FLOP: fwd 4*S*S*D per seq-head, bwd 2.5x)")
for S in [int(x) for x in a.seqs.split(",")]:
    B = max(1, a.tokens // S)
    T = B * S
    qkv = (torch.randn(T, 3 * H * D, device="cuda") * 0.5).to(torch.bfloat16)
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
    inv = 1.0 / (1000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    cos, sin = torch.cos(fr).cuda().contiguous(), torch.sin(fr).cuda().contiguous()
    cp, sp = (cos.data_ptr(), sin.data_ptr()) if a.rotary else (None, None)
    out = torch.empty(T, H * D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(H * T, device="cuda")
    dout = torch.randn_like(out)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(H * T, device="cuda")
    fwd = lambda: lib.cx_attn_varlen_fwd(qkv.data_ptr(), cu.data_ptr(), cp, sp, out.data_ptr(),
                                         lse.data_ptr(), B, H, T, S + a.max_seqlen_pad, 0.125, s)
    bwd = lambda: lib.cx_attn_varlen_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), cu.data_ptr(),
                                         cp, sp, delta.data_ptr(), dqkv.data_ptr(), B, H, T, S + a.max_seqlen_pad,
                                         0.125, s)
    if a.pdrop > 0:
        fwd = lambda: lib.cx_attn_varlen_dropout_fwd(qkv.data_ptr(), cu.data_ptr(), cp, sp, out.data_ptr(), lse.data_ptr(), B, H, T,
                                                     S + a.max_seqlen_pad, 0.125, a.pdrop, 1234, 0, 0, s)
        bwd = lambda: lib.cx_attn_varlen_dropout_bwd(dout.data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), cu.data_ptr(), cp, sp,
                                                     delta.data_ptr(), dqkv.data_ptr(), B, H, T, S + a.max_seqlen_pad, 0.125, a.pdrop, 1234, 0, 0, s)
    res = []
    for fn in (fwd, bwd):
        assert fn() == 0
        torch.cuda.synchronize()
        import time
        t_end = time.perf_counter() + (a.warm_seconds if not res else a.warm_seconds / 3)
        while time.perf_counter() < t_end:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / a.reps)
        if a.power > 0:
            from scripts.box_calibration import SmiSampler
            smp = SmiSampler(torch.cuda.current_device(), hz=20.0).start()
            t_end = time.perf_counter() + a.power
            while time.perf_counter() < t_end:
                for _ in range(4):
                    fn()
                torch.cuda.synchronize()
            smi = smp.stop()
            print(f"#   S = {S} {'fwd' if len(res) == 1 else 'bwd'}: {smi.get('mean_power_w', float('nan')):.0f} W, {smi.get('mean_sclk_mhz', float('nan')):.0f} MHz over {a.power:.1f} s of back-to-back launches")
    fl = 4.0 * S * S * D * B * H
    print(f"{S:5d} {B:5d} {res[0]:9.1f} {fl/res[0]/1e6:8.1f} {res[1]:9.1f} {2.5*fl/res[1]/1e6:8.1f}")
