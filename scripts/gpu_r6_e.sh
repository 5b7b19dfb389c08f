#!/bin/bash
# Round-6 call E: parity + microbench of the second-generation streaming backward kernels.
set -u
mkdir -p gpurun_out/r6e
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6e
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" > $O/tests_kernels.txt 2>&1; tail -3 $O/tests_kernels.txt
timeout 900 python -m pytest tests/test_dropout_gpu.py tests/test_vit_gpu.py tests/test_cfg3_gpu.py tests/test_shim_gpu.py tests/test_checkpoint_gpu.py -q > $O/tests_more.txt 2>&1; tail -3 $O/tests_more.txt
echo "## new kernels (no tables)" > $O/attn.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197,256,512,2048,8192 --rotary 0 >> $O/attn.txt 2>&1
echo "## round-1 backward kernels (cx_attn_set_bwd_long(0)), round-1 forward (cx_attn_set_fwd_long(0))" >> $O/attn.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197,256,512,2048,8192 --rotary 0 --bwd-long 0 --fwd-long 0 >> $O/attn.txt 2>&1
echo "## dropout 0.1: new | round 1" >> $O/attn.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197,2048 --rotary 0 --pdrop 0.1 >> $O/attn.txt 2>&1
grep -v amdgpu.ids $O/attn.txt
