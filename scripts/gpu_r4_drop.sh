#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dropout_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "dropout or attention" 2>&1 | tail -6
echo "== p = 0 (S <= 128 single-pass kernels)"; timeout 200 python scripts/attn_microbench.py --seqs 128 --tokens 262144 --reps 10 2>&1 | grep -v amdgpu
echo "== p = 0.1 single-pass <DROP>"; timeout 200 python scripts/attn_microbench.py --seqs 128 --tokens 262144 --reps 10 --pdrop 0.1 2>&1 | grep -v amdgpu
echo "== p = 0.1 general kernels"; timeout 200 python scripts/attn_microbench.py --seqs 128 --tokens 262144 --reps 10 --pdrop 0.1 --fwd-mode 0 --bwd-mode 0 2>&1 | grep -v amdgpu
