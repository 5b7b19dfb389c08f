#!/bin/bash
# Round 5, GPU call M: the GPU suite on the final code (dropout kernels changed last) + the three dropout legs of the bench
set -u
mkdir -p gpurun_out/r5m
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5m
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --steps 5 --only-config-legs cfg1,lit,clip > $O/legs.log 2>&1; grep "^{" $O/legs.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v['value'],1), round(v['ms_per_step'],2)) for k,v in d.items()})"
