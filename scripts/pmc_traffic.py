"""GEMM-family HBM traffic per launch from the two rocprofv3 PMC passes (scripts/gpu_round.sh, PMC=1).

usage: python scripts/pmc_traffic.py FETCH_summary.txt WRITE_summary.txt chunk > profiles/rN_pmc_gemm_traffic.json
Bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced stream
(LDS-DMA included) at 64 B, so the read side is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is.
The passes run bench.py at --global-batch == chunk, i.e. with the same per-launch GEMM sizes as the timed bench."""
import json
import re
import sys

GEMM = ("gemm_bf16_v6", "gemm_bf16_v5")  # v6 NT / SwiGLU / TN (v6tn) and the earlier generations


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
print(json.dumps({"grad_cache_chunk": chunk, "launches": calls,
                  "hbm_bytes_per_launch": (2 * kb_r + kb_w) * 1024 / max(1, calls),
                  "read_bytes_per_launch": 2 * kb_r * 1024 / max(1, calls), "write_bytes_per_launch": kb_w * 1024 / max(1, calls),
                  "correction": "read = 2 x FETCH_SIZE KB (gfx950), write = WRITE_SIZE KB", "per_kernel": per_kernel}, indent=1))
