#!/bin/bash
# per-shape L2-miss traffic of the wgrad kernel: FETCH_SIZE / WRITE_SIZE passes over the isolated launches
set -u
R=$PWD; O=$R/gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_wg_$ctr -o w -- python $R/scripts/gemm_microbench.py --chunk 2048 --reps 3 --only wgrad > $O/pmc_wg_$ctr.log 2>&1)
  python scripts/pmc_by_grid.py $(find $O/pmc_wg_$ctr -name "*counter_collection.csv" | head -1) $ctr v6tn > $O/wgrad_${ctr}_by_shape.txt 2>&1
  rm -rf $O/pmc_wg_$ctr
  cat $O/wgrad_${ctr}_by_shape.txt
done
grep -E "wgrad" $O/pmc_wg_FETCH_SIZE.log | head
