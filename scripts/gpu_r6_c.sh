#!/bin/bash
# Round-6 call C: parity + microbench of the attention kernels (S <= 256 backward without its scratch traffic, the long-sequence forward).
set -u
mkdir -p gpurun_out/r6c
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6c
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or act_bwd" > $O/tests_kernels.txt 2>&1; tail -3 $O/tests_kernels.txt
timeout 600 python -m pytest tests/test_dropout_gpu.py tests/test_vit_gpu.py tests/test_cfg3_gpu.py -q > $O/tests_drop_vit.txt 2>&1; tail -3 $O/tests_drop_vit.txt
echo "## S <= 256 kernels (no rotary)" > $O/attn.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197,256 --rotary 0 >> $O/attn.txt 2>&1
echo "## long sequences, no tables: new forward" >> $O/attn.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 512,2048,8192 --rotary 0 >> $O/attn.txt 2>&1
echo "## long sequences, no tables: round-1 forward (cx_attn_set_fwd_long(0))" >> $O/attn.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 512,2048,8192 --rotary 0 --fwd-long 0 >> $O/attn.txt 2>&1
grep -v amdgpu.ids $O/attn.txt
