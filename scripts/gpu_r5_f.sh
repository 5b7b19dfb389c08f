#!/bin/bash
# Round 5, GPU call F: what dropping the O read from the fused S <= 128 attention backward would buy (delta precomputed elsewhere).
set -u
mkdir -p gpurun_out/r5f
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5f
timeout 300 python scripts/lib_ab.py --libs base,din --cases attn_bwd_dpre --rounds 11 > $O/ab_attn_din.txt 2>&1; cat $O/ab_attn_din.txt
