#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 120 python scripts/gemm_v7_census.py > gpurun_out/r4/v7_census.txt 2>&1
cat gpurun_out/r4/v7_census.txt
timeout 400 python scripts/gemm_v7_ablate.py > gpurun_out/r4/v7_ablate.txt 2>&1
cat gpurun_out/r4/v7_ablate.txt
