#!/bin/bash
# round 4: first hardware run of the two-workgroups-per-CU GEMM (parity vs v6, then the A/B microbench)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "v7" 2>&1 | tail -15 > gpurun_out/r4/v7_tests.txt
cat gpurun_out/r4/v7_tests.txt
timeout 300 python scripts/gemm_v7_ab.py --chunk 2048 --rounds 5 --reps 8 > gpurun_out/r4/v7_ab.txt 2>&1
cat gpurun_out/r4/v7_ab.txt
