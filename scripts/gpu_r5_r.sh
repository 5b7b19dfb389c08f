#!/bin/bash
# Round 5, GPU call R: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the fused long-sequence backward vs the two-kernel backward; S = 512 / 1024 timing
set -u
mkdir -p gpurun_out/r5r
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r5r
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o p -- python $R/scripts/attn_bwd_long_ab.py --seq 2048 --tokens 131072 --timing-only 1 --rounds 1 --reps 2 > $O/pmc_$ctr.log 2>&1)
  f=$(find $O/pmc_$ctr -name "*counter_collection.csv" | head -1)
  [[ -n "$f" ]] && python scripts/pmc_summary.py $f $ctr | grep -E "^#|attn_" > $O/pmc_${ctr}_summary.txt
  rm -rf $O/pmc_$ctr
  cat $O/pmc_${ctr}_summary.txt | cut -c1-170
done
for S in 512 1024; do
  timeout 300 python scripts/attn_bwd_long_ab.py --seq $S --tokens 131072 --timing-only 1 > $O/s$S.txt 2>&1; grep -v "^/opt" $O/s$S.txt | tail -2
done
