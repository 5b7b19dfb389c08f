#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
R=$(pwd); O=$R/gpurun_out/r4; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c64 -o p -- python $R/bench.py --global-batch 2048 --chunk-size 64 --steps 2 --warmup 1 --no-extra-legs --no-cpu-baseline > $O/prof_c64.log 2>&1)
t=$(find $O/prof_c64 -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" > $O/kernel_summary_chunk64.txt 2>&1
rm -rf $O/prof_c64
grep "^{" $O/prof_c64.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs/s', d['value'], 'ms', d['ms_per_step'])"
head -24 $O/kernel_summary_chunk64.txt | cut -c1-150
