#!/bin/bash
# rocprofv3 kernel stats of the secondary config legs (cfg3, lit, clip): per-kernel time shares under profiles/
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3cfg; mkdir -p $O
for leg in cfg3 lit clip; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$leg -o p -- python $R/bench.py --steps 2 --only-config-legs $leg > $O/prof_$leg.log 2>&1)
  t=$(find $O/prof_$leg -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" 24 > $O/kernel_summary_$leg.txt 2>&1
  rm -rf $O/prof_$leg
  head -16 $O/kernel_summary_$leg.txt | cut -c1-170
done
