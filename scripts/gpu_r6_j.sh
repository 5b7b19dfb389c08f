#!/bin/bash
set -u
mkdir -p gpurun_out/r6j
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6j
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 600 python -m pytest tests/test_cfg3_gpu.py tests/test_vit_gpu.py tests/test_shim_gpu.py -q > $O/tests2.txt 2>&1; tail -2 $O/tests2.txt
for i in 1 2; do
echo "## pipelined full tiles (defer-max)" >> $O/attn.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 512,2048,8192 --rotary 0 >> $O/attn.txt 2>&1
echo "## first form (CX_ATTN_LONG_PIPE=0)" >> $O/attn.txt
CONTRASTORS_HIP_DEV_LIB=contrastors_amd/lib/variants/libcontrastors_hip_dev_nopipe.so timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 512,2048,8192 --rotary 0 >> $O/attn.txt 2>&1
done
grep -v amdgpu $O/attn.txt
