"""GEMM-family HBM traffic per launch from the two rocprofv3 PMC passes (scripts/gpu_round.sh, PMC=1).

usage: python scripts/pmc_traffic.py FETCH_summary.txt WRITE_summary.txt chunk > profiles/rN_pmc_gemm_traffic.json
Bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced stream
(LDS-DMA included) at 64 B, so the read side is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is.
The passes run bench.py at --global-batch == chunk, i.e. with the same per-launch GEMM sizes as the timed bench."""
import json
import re
import sys

GEMM = ("gemm_bf16_v6", "gemm_bf16_v5")  # v6 NT / SwiGLU / TN (v6tn) and the earlier generations


def algorithmic_bytes(chunk, seq=128, d=768, inner=3072):
    """ALGORITHMIC HBM bytes per launch of every GEMM class of the nomic-bert GradCache step (bf16 tensors; weights once;
    fused-epilogue side streams counted; fp32 split-K slabs of the wgrad launches left out: a few MB).  T = chunk * seq token
    rows per launch.  Returns ({kernel substring: (launches per layer-chunk, bytes per launch)}, family average).
    Derivation, X = T x d x 2 B, per layer and chunk:
      forward x 2 (pass 1 no save, pass 2 save):
        qkv       X + W(3d x d) + out 3X                         out_proj  X + W(d x d) + residual X + out X
        fc1+swiglu X + W(2I x d) + act T I 2 [+ gate T I 2 in pass 2; rounds 1-2 saved (y, gate): T 2I 2]
        fc2       act + W(d x I) + residual X + out X
      dgrad:
        fc2-dgrad + SwiGLU backward  X + W + (act, gate) read + d(y, gate) write    fc1-dgrad  d(y,gate) + W + add X + out X
        out_proj dgrad  X + W + out X                                               qkv dgrad  3X + W + add X + out X
      wgrad (natural layout, both operands streamed once):
        fc2  X + act      fc1  d(y,gate) + X      out_proj  X + X      qkv  3X + X"""
    T = chunk * seq
    X = T * d * 2.0
    A = T * inner * 2.0          # act
    YG = 2 * A                   # (y, gate) / its gradient
    Wqkv, Wo, W1, W2 = 3 * d * d * 2.0, d * d * 2.0, 2 * inner * d * 2.0, d * inner * 2.0
    plain = [X + Wqkv + 3 * X, X + Wo + X + X, A + W2 + X + X] * 2 + [YG + W1 + X + X, X + Wo + X, 3 * X + Wqkv + X + X]
    swiglu = [X + W1 + A, X + W1 + A + A]      # (pass 2 saves the gate alone, T I 2)
    swiglu_bwd = [X + W2 + YG + YG]            # (reads act + gate = the bytes of the (y, gate) pair it replaced)
    wgrad = [X + A, YG + X, X + X, 3 * X + X]
    classes = {"gemm_bf16_v6_kernel<0": plain, "gemm_bf16_v6_kernel<5": swiglu, "gemm_bf16_v6_kernel<6": swiglu_bwd,
               "gemm_bf16_v6tn_kernel": wgrad}   # <5> / <6>: the gate-save forms of <1> (SwiGLU) / <3> (SwiGLU backward)
    allb = sum(sum(v) for v in classes.values())
    n = sum(len(v) for v in classes.values())
    return {k: (len(v), sum(v) / len(v)) for k, v in classes.items()}, allb / n


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"\s*([0-9.]+) total\s+(\d+) calls\s+([0-9.]+) avg\s+(.*)", line)
        if m:
            out[m.group(4).strip()] = (float(m.group(1)), int(m.group(2)))
    return out


fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
chunk = int(sys.argv[3])
kb_r = kb_w = 0.0
calls = 0
per_kernel = {}
for k, (tot, n) in fetch.items():
    if any(g in k for g in GEMM):
        w = write.get(k, (0.0, n))[0]
        kb_r += tot
        kb_w += w
        calls += n
        per_kernel[k[:80]] = {"launches": n, "read_bytes_per_launch": 2 * tot * 1024 / n, "write_bytes_per_launch": w * 1024 / n}
alg, alg_avg = algorithmic_bytes(chunk)
for k, rec in per_kernel.items():
    for sub, (_, b) in alg.items():
        if sub in k:
            rec["algorithmic_bytes_per_launch"] = b
            rec["traffic_over_algorithmic"] = (rec["read_bytes_per_launch"] + rec["write_bytes_per_launch"]) / b
print(json.dumps({"grad_cache_chunk": chunk, "launches": calls,
                  "hbm_bytes_per_launch": (2 * kb_r + kb_w) * 1024 / max(1, calls),
                  "algorithmic_bytes_per_launch": alg_avg,
                  "traffic_over_algorithmic": (2 * kb_r + kb_w) * 1024 / max(1, calls) / alg_avg,
                  "read_bytes_per_launch": 2 * kb_r * 1024 / max(1, calls), "write_bytes_per_launch": kb_w * 1024 / max(1, calls),
                  "correction": "read = 2 x FETCH_SIZE KB (gfx950), write = WRITE_SIZE KB", "per_kernel": per_kernel}, indent=1))
