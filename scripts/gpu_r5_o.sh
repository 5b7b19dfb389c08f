#!/bin/bash
# Round 5, GPU call O: deferred WHOLE-LINE stores of the plain GEMM (DEFER mode 1: tile through the LDS staging rows into 128
# registers, 32 stores of 4 rows x 256 B in the next tile's K loop) against the staged epilogue, stores spread over 1 / 2 / 4 K-tiles
set -u
mkdir -p gpurun_out/r5o
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5o
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -x -k "deferred" > $O/tests_deferred.txt 2>&1; tail -2 $O/tests_deferred.txt
timeout 400 python scripts/lib_ab.py --libs base,d1,d2,d4 --cases qkv_fwd,out_dgrad --rounds 9 --reps 6 > $O/ab_defer.txt 2>&1; grep -v "^/opt" $O/ab_defer.txt | tail -4
