#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
R=$(pwd); O=$R/gpurun_out/r4; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg1 -o p -- python $R/bench.py --only-config-legs cfg1 --steps 10 > $O/prof_cfg1.log 2>&1)
t=$(find $O/prof_cfg1 -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" > $O/kernel_summary_cfg1.txt 2>&1
rm -rf $O/prof_cfg1
head -34 $O/kernel_summary_cfg1.txt | cut -c1-150
