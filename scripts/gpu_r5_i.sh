#!/bin/bash
# Round 5, GPU call I: the resident-schedule fallback gives the pooled arenas back before it runs two passes -- loss tests + the two-tenant test three times.
set -u
mkdir -p gpurun_out/r5i
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5i
timeout 600 python -m pytest tests/test_loss_gpu.py -q > $O/loss_tests.txt 2>&1; tail -2 $O/loss_tests.txt
for i in 1 2 3; do
  CX_TEST_TWO_TENANTS=1 timeout 300 python -m pytest tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape -q -s > $O/two_tenants_$i.txt 2>&1
  grep -E "passed|failed" $O/two_tenants_$i.txt | tail -1
  grep -E "^two tenants" $O/two_tenants_$i.txt | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln.split(':', 1)[1])
    print([[ (s['schedule'], s['fell_back'], round(s['peak_gb'])) for s in rk['steps']] for rk in r])"
done
