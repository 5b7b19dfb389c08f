#!/bin/bash
# start-skew experiment on the epilogue-heavy GEMM kernels: isolated launches per variant library
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
mkdir -p gpurun_out/r3skew
V=contrastors_amd/lib/variants
for v in base "$@"; do
  if [[ $v == base ]]; then lib=contrastors_amd/lib/libcontrastors_hip_dev.so; else lib=$V/libcontrastors_hip_dev_$v.so; fi
  echo "== $v"; CONTRASTORS_HIP_DEV_LIB=$lib timeout 300 python scripts/gemm_microbench.py --chunk 2048 --reps 8 2>&1 | tail -5
done | tee gpurun_out/r3skew/skew.txt
