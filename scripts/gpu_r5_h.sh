#!/bin/bash
# Round 5, GPU call H: register prefetch in the streaming attention backward kernels (S > 128): A/B at cfg 3's and the ViT's lengths.
# bpf = both kernels prefetch, 2 workgroups per CU; bpf1 = dQ kernel prefetches at 3 workgroups per CU (6 spills); bpf2 = dK/dV kernel prefetches at 2 per CU
set -u
mkdir -p gpurun_out/r5h
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5h
for S in 2048 512 197; do
  timeout 300 python scripts/lib_ab.py --libs base,bpf,bpf1,bpf2 --cases attn_bwd --seq $S --rounds 7 --reps 3 > $O/ab_bpf_$S.txt 2>&1; grep -v "^/opt" $O/ab_bpf_$S.txt
done
