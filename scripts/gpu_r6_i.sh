#!/bin/bash
set -u
mkdir -p gpurun_out/r6i
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6i
CX_BENCH_CHUNK=4096 timeout 1500 python bench.py --steps 5 --warmup 2 > $O/bench_4096.log 2>&1
grep '^{' $O/bench_4096.log | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read())
print("headline", round(d["value"],1), d["config"], "peak", d.get("peak_hbm_gb"))
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if isinstance(v,(int,float)) and any(t in k for t in ("cfg","lit","clip","weak","resident","auto","dropin","chunk64","vs_box","like_for"))})
'
tail -5 $O/bench_4096.log | cut -c1-200
