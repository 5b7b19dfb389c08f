#!/bin/bash
# Round 5, GPU call H: streaming attention backward variants (S > 128), A/B at cfg 3's and the ViT's lengths.
# base = shipped (CX_ATTN_BWD_PF 2: dK/dV kernel prefetches the next chunk, 2 workgroups per CU); db10 = + double-buffered chunk tiles (one barrier per chunk)
set -u
mkdir -p gpurun_out/r5h
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5h
for S in 2048 512 197; do
  timeout 300 python scripts/lib_ab.py --libs base,${1:-db10} --cases attn_bwd --seq $S --rounds 7 --reps 3 > $O/ab_${1:-db10}_$S.txt 2>&1; grep -v "^/opt" $O/ab_${1:-db10}_$S.txt
done
