#!/bin/bash
# Round 5, GPU call L: dropout form of the fused S <= 128 attention backward with its keep bits drawn ahead of the S / dP products (40 instead of 57 spills)
set -u
mkdir -p gpurun_out/r5l
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5l
for rot in 1 0; do
  timeout 300 python scripts/lib_ab.py --libs base,de --cases attn_bwd_drop --rotary $rot --rounds 9 > $O/ab_de_rot$rot.txt 2>&1; grep -v "^/opt" $O/ab_de_rot$rot.txt
done
timeout 300 python scripts/lib_ab.py --libs base,de --cases attn_bwd_drop --seq 64 --rotary 0 --rounds 9 > $O/ab_de_s64.txt 2>&1; grep -v "^/opt" $O/ab_de_s64.txt
