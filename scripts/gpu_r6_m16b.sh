#!/bin/bash
# 16x16x32 main loop: the whole GPU suite, the ablation of the NT kernel, the bench line.
set -u
out=gpurun_out/r6m; mkdir -p $out
python scripts/gemm_ablate.py > $out/ablate_m16.txt 2>&1; echo "ablate rc=$?"
timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests_all.txt 2>&1; echo "tests rc=$?"; tail -n 3 $out/tests_all.txt
python bench.py > $out/bench_m16.json 2> $out/bench_m16.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6m/bench_m16.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'vs_box_blas', 'like_for_like', 'like_for_like_short_k', 'like_for_like_long_k', 'cfg1_pairs_s', 'cfg3_query_examples_s', 'lit_pairs_s', 'clip_pairs_s')})
print(d.get('roofline'), d.get('box'))
PY
