#!/bin/bash
# Round 5, GPU call B: the whole GPU suite, same-box A/B of the fused S <= 128 attention backward variants, XCD-grid x power sweep
# of the GEMMs with FETCH_SIZE per grid.
set -u
mkdir -p gpurun_out/r5b
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r5b
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape > $O/gpu_tests.txt 2>&1; tail -15 $O/gpu_tests.txt | cut -c1-300
CX_TEST_TWO_TENANTS=1 timeout 400 python -m pytest tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape -x -q -s > $O/two_tenants.txt 2>&1; grep -E "two tenants|passed|failed|Error" $O/two_tenants.txt | cut -c1-2500
timeout 400 python scripts/lib_ab.py --libs base,rot,rm,rmrot --cases attn_bwd,attn_bwd_ragged --rounds 9 > $O/ab_attn_bwd.txt 2>&1; cat $O/ab_attn_bwd.txt
timeout 300 python scripts/gemm_grid_power_sweep.py > $O/grid_power.txt 2>&1; cat $O/grid_power.txt
for gn in 1 2 4 8; do
  for c in fc1_swiglu_save qkv_fwd; do
    (cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${c}_$gn -o p -- python $R/scripts/gemm_grid_power_sweep.py --pmc $c $gn > $O/pmc_${c}_$gn.log 2>&1)
    f=$(find $O/pmc_${c}_$gn -name "*counter_collection.csv" | head -1)
    [[ -n "$f" ]] && echo "$c gn=$gn $(python scripts/pmc_summary.py $f FETCH_SIZE | grep gemm_bf16_v6 | head -1)" >> $O/grid_fetch.txt
    rm -rf $O/pmc_${c}_$gn
  done
done
cat $O/grid_fetch.txt
