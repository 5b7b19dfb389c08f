#!/bin/bash
# attention backward A/B: store_unrotated_rows with the cos / sin fetched one column group ahead (default) vs as in round 2
set -u
mkdir -p gpurun_out/r3attn
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r3attn
V=contrastors_amd/lib/variants
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_dropout_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 1 2; do
echo "rot ahead (default)"; timeout 200 python scripts/attn_microbench.py --tokens 262144 --seqs 128,512,2048 --reps 10 2>&1 | tail -3
echo "rot0"; CONTRASTORS_HIP_DEV_LIB=$V/libcontrastors_hip_dev_rot0.so timeout 200 python scripts/attn_microbench.py --tokens 262144 --seqs 128,512,2048 --reps 10 2>&1 | tail -3
done | tee $O/ab_rot.txt
