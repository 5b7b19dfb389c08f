#!/bin/bash
# Round 5, GPU call E: residual-preload variant of the plain GEMM epilogue (A/B + bit-identity), then the whole step with it.
set -u
mkdir -p gpurun_out/r5e
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5e
timeout 500 python scripts/lib_ab.py --libs base,rpre --cases out_fwd_res,fc2_fwd_res,qkv_dgrad_res,fc1_dgrad_res,qkv_fwd --rounds 9 > $O/ab_res_pre.txt 2>&1; cat $O/ab_res_pre.txt
L=contrastors_amd/lib
cp $L/libcontrastors_hip.so /tmp/base.so
for v in base rpre base rpre; do
  if [[ $v == base ]]; then cp /tmp/base.so $L/libcontrastors_hip.so; else cp $L/variants/libcontrastors_hip_$v.so $L/libcontrastors_hip.so; fi
  timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --no-calibration > $O/bench_$v.log 2>&1
  echo "$v: $(grep '^{' $O/bench_$v.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["achieved"], d["step_ms"]["median"])')" | tee -a $O/step_ab.txt
done
cp /tmp/base.so $L/libcontrastors_hip.so
