#!/bin/bash
# Round-6 call F: the whole GPU suite after the tidy + config legs + attention microbench (new | round-1 kernels).
set -u
mkdir -p gpurun_out/r6f
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6f
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
timeout 900 python bench.py --steps 3 --warmup 1 --only-config-legs cfg1,lit,clip,cfg3 > $O/legs.log 2>&1
grep "^{" $O/legs.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('cfg1','cfg3','lit','clip'):
    r=d.get(k,{})
    print(k, round(r.get('value',0),1), r.get('unit'), round(r.get('ms_per_step',0),1),'ms', (r.get('roofline') or {}).get('frac'), (r.get('selective_checkpointing') or {}).get('value'))
"
