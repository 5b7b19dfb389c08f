#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
for f in 0 1; do
  echo "== flags $f"
  timeout 300 python scripts/gemm_v7_ab.py --chunk 2048 --rounds 3 --reps 8 --flags $f 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/v7_ab_flags$f.txt
done
