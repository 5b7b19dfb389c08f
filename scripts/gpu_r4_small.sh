#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -4
timeout 600 python bench.py --only-config-legs cfg1,cfg1_nodrop --steps 10 2>&1 | tail -1 > gpurun_out/r4/legs_cfg1_v7.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/legs_cfg1_v7.json"))
for k,v in d.items(): print(k, round(v["value"],1), v["unit"], "ms", round(v["ms_per_step"],2))
PY
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config-legs 2>&1 | tail -1 > gpurun_out/r4/bench_noconfig.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/bench_noconfig.json"))
print("headline", round(d["value"],1), "ms", round(d["ms_per_step"],1), "frac", round(d["roofline"]["frac"],4), "whole", round(d["roofline"]["whole_step_frac_of_mfma_peak"],4))
for k in ("weak","resident","dropin_chunk64"):
    v=d.get(k); print(k, v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in("value","ms_per_step","exact_chunk64")})
PY
