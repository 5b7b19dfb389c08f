#!/bin/bash
# Round-6 call D: phase trace of the S <= 256 backward + the config legs (cfg1 / lit / clip / cfg3) with the round's kernels.
set -u
mkdir -p gpurun_out/r6d
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6d
CONTRASTORS_HIP_DEV_LIB=contrastors_amd/lib/variants/libcontrastors_hip_dev_attntrace.so timeout 200 python scripts/attn_trace_s256.py 197 > $O/trace_s256_197.txt 2>&1
CONTRASTORS_HIP_DEV_LIB=contrastors_amd/lib/variants/libcontrastors_hip_dev_attntrace.so timeout 200 python scripts/attn_trace_s256.py 256 > $O/trace_s256_256.txt 2>&1
grep -v amdgpu.ids $O/trace_s256_197.txt; grep -v amdgpu.ids $O/trace_s256_256.txt | head -4
timeout 900 python bench.py --steps 3 --warmup 1 --only-config-legs cfg1,lit,clip,cfg3 > $O/legs.log 2>&1
grep "^{" $O/legs.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if isinstance(v,(int,float)) and any(t in k for t in ('cfg1','cfg3','lit','clip')): print(k, round(v,4))
"
