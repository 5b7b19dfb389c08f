#!/bin/bash
# Round-6 call B: parity of the new kernels (S <= 256 attention, fused activation backward) + their microbenchmarks.
set -u
mkdir -p gpurun_out/r6b
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6b
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or act_bwd or bias_gelu" > $O/tests_kernels.txt 2>&1; tail -5 $O/tests_kernels.txt
timeout 600 python -m pytest tests/test_dropout_gpu.py tests/test_vit_gpu.py -q > $O/tests_drop_vit.txt 2>&1; tail -5 $O/tests_drop_vit.txt
for pad in 0 100; do
  echo "## max_seqlen pad $pad (100: general streaming kernels), no rotary" >> $O/attn_s197.txt
  timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197,256,160 --rotary 0 --max-seqlen-pad $pad >> $O/attn_s197.txt 2>&1
done
echo "## dropout 0.1, pad 0 / 100" >> $O/attn_s197.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197 --rotary 0 --pdrop 0.1 >> $O/attn_s197.txt 2>&1
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197 --rotary 0 --pdrop 0.1 --max-seqlen-pad 100 >> $O/attn_s197.txt 2>&1
grep -v amdgpu.ids $O/attn_s197.txt
