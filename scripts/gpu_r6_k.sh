#!/bin/bash
set -u
mkdir -p gpurun_out/r6k
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6k
SECONDS=0; python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "rc=$? wall ${SECONDS}s"
grep '^{' $O/bench_default.log | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read())
print("headline", round(d["value"],1), "steps", d["steps"], "warmup", d["warmup"], d["config"]["grad_cache_chunk"], "resident", d.get("resident_pairs_s"), (d.get("resident") or {}).get("peak_hbm_gb") if isinstance(d.get("resident"),dict) else d.get("resident"))
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if isinstance(v,(int,float)) and any(t in k for t in ("cfg","lit","clip","weak","auto","dropin","chunk64"))})
'
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
