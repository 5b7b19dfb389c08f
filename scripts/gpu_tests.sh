#!/bin/bash
# full GPU test suite + smoke; usage: bash scripts/gpu_tests.sh [pytest args]
set -u
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
rm -f gpurun_out/kernel_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 "$@" > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log
grep -E "passed|failed|error|Error|assert " gpurun_out/pytest.log | tail -40
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
