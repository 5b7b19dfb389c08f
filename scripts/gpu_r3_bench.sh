#!/bin/bash
# bench line on this box (+ optional variant library): usage  bash scripts/gpu_r3_bench.sh [bench args]
set -u
mkdir -p gpurun_out/r3bench
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1500 python bench.py "$@" > gpurun_out/r3bench/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/r3bench/bench.log
tail -2 gpurun_out/r3bench/bench.log | cut -c1-6000
