#!/bin/bash
# The 16x16x32 main loop (gemm_bf16_v6.hip): parity first, then the product against the 32x32x16 build of the commit before
# (contrastors_amd/lib/variants/libcontrastors_hip_m32.so) through scripts/v6_vs_vendor.py time, alternating.
set -u
out=gpurun_out/r6m; mkdir -p $out
V=contrastors_amd/lib/variants
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or wgrad or swiglu or gelu or act_bwd or linear" > $out/tests_gemm.txt 2>&1
echo "gemm tests rc=$?"; tail -n 5 $out/tests_gemm.txt
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_m32.so python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_m32_$r.txt 2>&1
  python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_m16real_$r.txt 2>&1
done
tail -n 10 $out/time_m32_2.txt $out/time_m16real_2.txt
