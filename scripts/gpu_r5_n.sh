#!/bin/bash
# Round 5, GPU call N: plain streaming dK/dV kernel at two workgroups per CU WITHOUT the prefetch (no spills at 238 registers) against three per CU (7 spills at 168)
set -u
mkdir -p gpurun_out/r5n
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5n
for S in 2048 197; do
  timeout 300 python scripts/lib_ab.py --libs base,w2 --cases attn_bwd --seq $S --rotary 0 --chunk 1024 --rounds 7 --reps 3 > $O/ab_w2_$S.txt 2>&1; grep -v "^/opt" $O/ab_w2_$S.txt | tail -1
done
