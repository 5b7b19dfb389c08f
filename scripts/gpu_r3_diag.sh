#!/bin/bash
# Round-3 diagnostics call: phase timers of the fused attention backward, per-parameter parity tables, kernel baselines.
set -u
mkdir -p gpurun_out/r3diag
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r3diag
V=contrastors_amd/lib/variants
CONTRASTORS_HIP_DEV_LIB=$V/libcontrastors_hip_dev_attntrace.so timeout 300 python scripts/attn_trace.py > $O/attn_trace.txt 2>&1
cat $O/attn_trace.txt | tail -16
rm -f gpurun_out/kernel_report.jsonl
timeout 600 python -m pytest tests/test_loss_gpu.py::test_grad_cache_loss_equals_full_batch_oracle tests/test_vit_gpu.py::test_vit_b16_vs_oracle tests/test_engine_gpu.py -m gpu -q > $O/pytest_parity.log 2>&1
tail -3 $O/pytest_parity.log
cp gpurun_out/kernel_report.jsonl $O/parity_report.jsonl
timeout 300 python scripts/attn_microbench.py --tokens 262144 > $O/attn_microbench.txt 2>&1; cat $O/attn_microbench.txt
timeout 300 python scripts/gemm_microbench.py --chunk 2048 --reps 6 > $O/gemm_microbench.txt 2>&1; tail -18 $O/gemm_microbench.txt
