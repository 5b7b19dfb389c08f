#!/bin/bash
# Round 5, GPU call Q: ablations of the fused long-sequence backward (fd1 no scratch traffic, fd2 no dQ product either, fd4 scratch written, never read)
set -u
mkdir -p gpurun_out/r5q
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5q
timeout 300 python scripts/attn_bwd_long_ab.py --seq 2048 --tokens 131072 --libs dev,fd1,fd2,fd4 --timing-only 1 > $O/s2048_ablate.txt 2>&1; grep -v "^/opt" $O/s2048_ablate.txt | tail -6
