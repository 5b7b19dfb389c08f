#!/bin/bash
# Round 5, GPU call P: the fused single-owner long-sequence attention backward against the two-kernel backward
set -u
mkdir -p gpurun_out/r5p
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5p
timeout 300 python scripts/attn_bwd_long_ab.py --seq 512 --tokens 8192 --ragged 1 --libs dev,fl2 --rounds 3 > $O/small_ragged.txt 2>&1; grep -v "^/opt" $O/small_ragged.txt | tail -12
timeout 300 python scripts/attn_bwd_long_ab.py --seq 2048 --tokens 131072 --libs dev,fl2 > $O/s2048.txt 2>&1; grep -v "^/opt" $O/s2048.txt | tail -9
