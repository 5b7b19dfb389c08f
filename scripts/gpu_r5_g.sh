#!/bin/bash
# Round 5, GPU call G: the calibration with its second (LDS-fed) probe + a short bench line -- run on as many boxes as calls land on.
set -u
mkdir -p gpurun_out/r5g
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5g; tag=$(date +%H%M%S)
timeout 120 python scripts/box_calibration.py > $O/cal_$tag.json 2>$O/cal_$tag.err; cut -c1-900 $O/cal_$tag.json
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_$tag.log 2>&1; grep "^{" $O/bench_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('box_') or k.startswith('frac_of') or k in ('value','vs_box_blas')}, round(d['roofline']['achieved'],1))" | tee $O/line_$tag.txt
