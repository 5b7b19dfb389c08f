"""Time cx_gemm_bf16_nt on the encoder's GEMM shapes (HIP events, random bf16 data).  usage:
   python scripts/gemm_microbench.py [--variant 2] [--chunk 64] [--reps 20] [--shapes fwd|all]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd import _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=6)
ap.add_argument("--glds", type=int, default=1)
ap.add_argument("--chunk", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", type=str, default="")
ap.add_argument("--dbg", type=int, default=0)
ap.add_argument("--tn", type=int, default=1)
ap.add_argument("--zeros", type=int, default=0, help="all-zero operands: no switching power, clocks stay high")
ap.add_argument("--shape", type=str, action="append", default=[], help="M,N,K (repeatable)")
ap.add_argument("--warm-seconds", type=float, default=0.5, help="back-to-back launches of each shape before it is timed: the package at its sustained "
                "clock under the power cap, not at its boost clock (VERDICT r5 weak item 5: the first row of r5_microbench_gemm2048.txt read 892 TF "
                "for a launch that sustains 1146)")
a = ap.parse_args()
lib = _C.dev_lib()  # variant / ablation switches live in the dev library
lib.cx_gemm_set_variant(a.variant)
lib.cx_gemm_set_glds(a.glds)
lib.cx_gemm_set_debug(a.dbg)
lib.cx_gemm_v6_ablate(0)
T = a.chunk * 128
shapes = {  # name: (M, N, K)
    "qkv_fwd": (T, 2304, 768), "out_fwd": (T, 768, 768), "fc1_fwd": (T, 6144, 768), "fc2_fwd": (T, 768, 3072),
    "fc1_dgrad": (T, 768, 6144), "fc2_dgrad": (T, 3072, 768), "qkv_dgrad": (T, 768, 2304),
    "qkv_wgrad": (2304, 768, T), "fc1_wgrad": (6144, 768, T), "fc2_wgrad": (768, 3072, T), "out_wgrad": (768, 768, T),
}
if a.shape:
    shapes = {f"custom{i}": tuple(int(v) for v in sh.split(",")) for i, sh in enumerate(a.shape)}
dev = "cuda"
s = torch.cuda.current_stream().cuda_stream
tot_f = tot_t = 0.0
for name, (M, N, K) in shapes.items():
    if a.only and a.only not in name:
        continue
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    if a.zeros:
        x.zero_(); w.zero_()
    wg = name.endswith("wgrad")
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if wg else torch.bfloat16)
    ws = torch.empty(8 * 6144 * 768 if wg else 1, device=dev)

    if wg and a.tn:
        dyt, at = x.T.contiguous(), w.T.contiguous()  # (T, O), (T, I)

    def run():
        if wg and a.tn:
            return lib.cx_gemm_bf16_tn_accum(dyt.data_ptr(), at.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), K, M,
                                             N, M, N, s)
        if wg:
            return lib.cx_gemm_bf16_nt_accum(x.data_ptr(), w.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), M, N,
                                             K, K, K, s)
        return lib.cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, K, K, K, N, 0, 1, 1.0, s)

    for _ in range(3):
        assert run() == 0
    import time
    t_end = time.perf_counter() + a.warm_seconds
    while time.perf_counter() < t_end:
        for _ in range(8):
            run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    fl = 2.0 * M * N * K
    tot_f += fl
    tot_t += us
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:6d}  {us:8.1f} us  {fl/us/1e6:7.1f} TF")
print(f"sum: {tot_t:.1f} us  {tot_f/tot_t/1e6:.1f} TF  (variant {a.variant}, chunk {a.chunk})")

# ---- the fused epilogue forms the engine actually launches (nomic-bert-2048 block) ----------------------------------
if not a.shape and not a.only:
    d, I = 768, 3072
    x = torch.randn(T, d, device=dev).bfloat16()
    res = torch.randn(T, d, device=dev).bfloat16()
    w1 = (torch.randn(2 * I, d, device=dev) * 0.05).bfloat16()
    w2 = (torch.randn(d, I, device=dev) * 0.05).bfloat16()
    w2t = w2.T.contiguous()
    wo = (torch.randn(d, d, device=dev) * 0.05).bfloat16()
    yg = torch.randn(T, 2 * I, device=dev).bfloat16()
    dyg = torch.empty_like(yg)
    act = torch.randn(T, I, device=dev).bfloat16()
    out = torch.empty(T, d, device=dev, dtype=torch.bfloat16)
    gsave = torch.randn(T, I, device=dev).bfloat16()
    fused = {
        "fc1+swiglu (save yg)": (2.0 * T * 2 * I * d, lambda: lib.cx_gemm_bf16_swiglu(x.data_ptr(), w1.data_ptr(), yg.data_ptr(), act.data_ptr(), T, I, d, d, d, 2 * I, I, s)),
        "fc1+swiglu (no save)": (2.0 * T * 2 * I * d, lambda: lib.cx_gemm_bf16_swiglu(x.data_ptr(), w1.data_ptr(), None, act.data_ptr(), T, I, d, d, d, 2 * I, I, s)),
        "fc2 dgrad+swiglu bwd": (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_swiglu_bwd(x.data_ptr(), w2t.data_ptr(), yg.data_ptr(), dyg.data_ptr(), T, I, d, d, d, 2 * I, s)),
        "fc1+swiglu (save gate)": (2.0 * T * 2 * I * d, lambda: lib.cx_gemm_bf16_swiglu_gate(x.data_ptr(), w1.data_ptr(), gsave.data_ptr(), act.data_ptr(), T, I, d, d, d, I, I, s)),
        "fc2 dgrad+swiglu bwd (act, gate)": (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_swiglu_bwd_gate(x.data_ptr(), w2t.data_ptr(), act.data_ptr(), gsave.data_ptr(), dyg.data_ptr(), T, I, d, d, d, I, 2 * I, s)),
        "fc2 fwd + residual": (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_nt_residual(act.data_ptr(), w2.data_ptr(), out.data_ptr(), None, res.data_ptr(), T, d, I, I, I, d, d, s)),
        "out_proj fwd + residual": (2.0 * T * d * d, lambda: lib.cx_gemm_bf16_nt_residual(x.data_ptr(), wo.data_ptr(), out.data_ptr(), None, res.data_ptr(), T, d, d, d, d, d, d, s)),
    }
    # the plain (GELU) MLP's backward pair (BERT-base / ViT towers): fused kernel of round 6 against the two kernels it replaces
    dpre = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
    dact = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
    dbias = torch.zeros(I, device=dev)
    wsb = torch.empty(((T + 127) // 128) * I, device=dev)

    def two_kernels():
        r = lib.cx_gemm_bf16_nt(x.data_ptr(), w2t.data_ptr(), dact.data_ptr(), None, T, I, d, d, d, I, 0, 1, 1.0, s)
        return r or lib.cx_bias_act_bwd_colsum(dact.data_ptr(), act.data_ptr(), None, dpre.data_ptr(), dbias.data_ptr(), T, I, 0, s)

    fused["fc2 dgrad + gelu bwd + db (fused)"] = (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_act_bwd(
        x.data_ptr(), w2t.data_ptr(), act.data_ptr(), dpre.data_ptr(), dbias.data_ptr(), wsb.data_ptr(), wsb.numel(), T, I, d, d, d, I, I, 0, s))
    fused["fc2 dgrad, then gelu bwd + db"] = (2.0 * T * I * d, two_kernels)
    # the plain MLP's forward: fc1 + bias + erf-GELU (BERT-base / ViT towers), pre-activation kept (training) or not (no-grad pass)
    wg = (torch.randn(I, d, device=dev) * 0.05).bfloat16()
    bg = torch.randn(I, device=dev) * 0.1
    pre = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
    fused["fc1 + bias + gelu (save pre)"] = (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_bias_gelu(x.data_ptr(), wg.data_ptr(), bg.data_ptr(), pre.data_ptr(), dact.data_ptr(), T, I, d, d, d, I, I, s))
    fused["fc1 + bias + gelu (no save)"] = (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_bias_gelu(x.data_ptr(), wg.data_ptr(), bg.data_ptr(), None, dact.data_ptr(), T, I, d, d, d, I, I, s))
    fused["fc1 plain (same shape)"] = (2.0 * T * I * d, lambda: lib.cx_gemm_bf16_nt(x.data_ptr(), wg.data_ptr(), dact.data_ptr(), None, T, I, d, d, d, I, 0, 1, 1.0, s))
    for name, (fl, run) in fused.items():
        for _ in range(3):
            assert run() == 0, name
        t_end = time.perf_counter() + a.warm_seconds
        while time.perf_counter() < t_end:
            for _ in range(8):
                run()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        print(f"{name:34s} {us:8.1f} us  {fl/us/1e6:7.1f} TF")
