"""Does an HBM-bound kernel co-run with the one-wave-per-SIMD GEMM on the same CUs?  (scripts/, measurement only)

The GEMM family holds 256 AGPRs + 124..254 VGPRs and all 160 KiB of LDS per workgroup; a LayerNorm wave needs <= 96
registers and no LDS, so the register file has room for it next to the wgrad kernel (380 registers) but not next to the
plain NT kernel (472).  This probe times  [GEMM x r] and [LN x r]  back to back on one stream, then on two streams,
and prints how much of the LayerNorm time disappears under the GEMM.  usage: python scripts/concurrency_probe.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

lib = _C.dev_lib()
dev = "cuda"
T, d, I = 131072, 768, 3072
reps = 6


def ev():
    return torch.cuda.Event(enable_timing=True)


x = torch.randn(T, d, device=dev).bfloat16()
dy = torch.randn(T, 3 * d, device=dev).bfloat16()
gw = torch.zeros(3 * d, d, device=dev)
ws = torch.empty(8 * 6144 * 768, device=dev)
w_nt = (torch.randn(3 * d, d, device=dev) * 0.05).bfloat16()
out_nt = torch.empty(T, 3 * d, device=dev, dtype=torch.bfloat16)
yg = torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16)
act = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
w_fc1 = (torch.randn(2 * I, d, device=dev) * 0.05).bfloat16()
gamma = torch.ones(d, device=dev)
beta = torch.zeros(d, device=dev)
ln_in = torch.randn(T, d, device=dev).bfloat16()
ln_out = torch.empty_like(ln_in)
mean = torch.empty(T, device=dev)
rstd = torch.empty(T, device=dev)


def gemm_tn(s):
    return lib.cx_gemm_bf16_tn_accum(dy.data_ptr(), x.data_ptr(), gw.data_ptr(), ws.data_ptr(), ws.numel(), T, 3 * d, d,
                                     3 * d, d, s)


def gemm_nt(s):
    return lib.cx_gemm_bf16_nt(x.data_ptr(), w_nt.data_ptr(), out_nt.data_ptr(), None, T, 3 * d, d, d, d, 3 * d, 0, 1,
                               1.0, s)


def gemm_swiglu(s):
    return lib.cx_gemm_bf16_swiglu(x.data_ptr(), w_fc1.data_ptr(), None, act.data_ptr(), T, I, d, d, d, 2 * I, I, s)


def ln(s):
    return lib.cx_layernorm_fwd(ln_in.data_ptr(), None, gamma.data_ptr(), beta.data_ptr(), ln_out.data_ptr(), None,
                                mean.data_ptr(), rstd.data_ptr(), T, d, 1e-12, s)


def timed(fn_a, fn_b, two_streams):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    sa.wait_event(e0)
    sb.wait_event(e0)
    with torch.cuda.stream(sa):
        for _ in range(reps):
            assert fn_a(sa.cuda_stream) == 0
    with torch.cuda.stream(sb if two_streams else sa):
        st = (sb if two_streams else sa).cuda_stream
        for _ in range(reps * 4):
            assert fn_b(st) == 0
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def alone(fn, n):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        fn(s)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(n):
        fn(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


t_ln = alone(ln, 12)
print(f"layernorm fwd alone: {t_ln:.1f} us ({3 * T * d * 2 / t_ln / 1e6:.2f} TB/s)")
for name, fn in (("wgrad v6tn (380 regs)", gemm_tn), ("fc1+swiglu v6<1> (400 regs)", gemm_swiglu), ("nt v6<0> (472 regs)", gemm_nt)):
    t_g = alone(fn, 6)
    serial = timed(fn, ln, False)
    conc = timed(fn, ln, True)
    hidden = (serial - conc) / (4 * reps * t_ln)
    print(f"{name:30s} alone {t_g:8.1f} us | {reps} gemm + {4 * reps} ln: one stream {serial:9.1f} us, two streams {conc:9.1f} us "
          f"-> {100 * hidden:5.1f} % of the LayerNorm time hidden")
