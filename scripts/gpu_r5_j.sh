#!/bin/bash
# Round 5, GPU call J: the whole GPU suite on the final code + rocprofv3 kernel tables of the cfg 3 leg (literal and selective checkpointing).
set -u
mkdir -p gpurun_out/r5j
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r5j
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for leg in cfg3 cfg3_selective; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$leg -o p -- python $R/bench.py --steps 2 --only-config-legs $leg > $O/prof_$leg.log 2>&1)
  t=$(find $O/prof_$leg -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" 24 > $O/kernel_summary_$leg.txt 2>&1
  rm -rf $O/prof_$leg
  head -12 $O/kernel_summary_$leg.txt | cut -c1-140
done
