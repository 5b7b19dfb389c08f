#!/bin/bash
# Round 5, GPU call D: LDS-DMA prefetch variant of the fused S <= 128 attention backward (A/B + parity), dropout after the Philox change.
set -u
mkdir -p gpurun_out/r5d
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5d
timeout 300 python scripts/lib_ab.py --libs base,dma --cases attn_bwd,attn_bwd_ragged --rounds 9 > $O/ab_attn_dma.txt 2>&1; cat $O/ab_attn_dma.txt
timeout 600 python -m pytest tests/test_dropout_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -k "drop or attn or attention or engine" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 128,197,512,2048 --pdrop 0.1 > $O/attn_p01.txt 2>&1; cat $O/attn_p01.txt | tail -5
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 128,197,512,2048 > $O/attn_p0.txt 2>&1; cat $O/attn_p0.txt | tail -5
CX_TEST_TWO_TENANTS=1 timeout 400 python -m pytest tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape -x -q -s > $O/two_tenants.txt 2>&1; grep -E "two tenants|passed|failed|Error" $O/two_tenants.txt | cut -c1-3000
