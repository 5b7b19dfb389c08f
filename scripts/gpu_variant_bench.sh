#!/bin/bash
# A/B a set of library variants (scripts/build_variant.py) on the whole step and on isolated GEMM launches: swaps the
# product .so on the (scratch) GPU box.  usage: [CAL=1] bash scripts/gpu_variant_bench.sh base nt1 nt3
set -u
mkdir -p gpurun_out/variants
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
L=contrastors_amd/lib
cp $L/libcontrastors_hip.so /tmp/base.so
for v in "$@"; do
  if [[ $v == base ]]; then cp /tmp/base.so $L/libcontrastors_hip.so; else cp $L/variants/libcontrastors_hip_$v.so $L/libcontrastors_hip.so; fi
  if [[ "${CAL:-0}" == 1 ]]; then timeout 300 python scripts/blas_calibration.py 2>&1 | grep -E "fwd|grad" | sed "s/^/$v  /" | tee gpurun_out/variants/cal_$v.txt; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > gpurun_out/variants/bench_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/variants/bench_$v.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["achieved"])')"
done
cp /tmp/base.so $L/libcontrastors_hip.so
