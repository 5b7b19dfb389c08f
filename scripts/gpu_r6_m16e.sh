#!/bin/bash
# Spread LDS-DMA issue (one DMA per CX_V6_DMA_SPREAD MFMAs instead of back to back): ablation cycles and timing, spread 4 (built default) / 2 (sp2) / 1 (m16a)
set -u
out=gpurun_out/r6m; mkdir -p $out
V=contrastors_amd/lib/variants
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or wgrad" > $out/tests_gemm_e.txt 2>&1; echo "gemm tests rc=$?"; tail -n 1 $out/tests_gemm_e.txt
python scripts/gemm_ablate.py > $out/ablate_sp4.txt 2>&1
CONTRASTORS_HIP_DEV_LIB=$V/libcontrastors_hip_dev_sp2.so python scripts/gemm_ablate.py > $out/ablate_sp2.txt 2>&1
for f in sp4 sp2; do echo "== $f"; grep -v amdgpu.ids $out/ablate_$f.txt | cut -c1-170 | grep "^==\|full kernel again\|no epilogue\|no DMA  \|MFMA only, no"; done
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_m16a.so python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_e_sp1_$r.txt 2>&1
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_sp2.so python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_e_sp2_$r.txt 2>&1
  python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_e_sp4_$r.txt 2>&1
done
tail -n 9 $out/time_e_sp1_2.txt $out/time_e_sp2_2.txt $out/time_e_sp4_2.txt
