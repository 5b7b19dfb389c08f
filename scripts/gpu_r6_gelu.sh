#!/bin/bash
# fast path of the fc1 + bias + GELU epilogue (built) against the library before it (variant pre): parity, microbenchmark rows, the GELU legs
set -u
out=gpurun_out/r6m; mkdir -p $out
V=contrastors_amd/lib/variants
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_vit_gpu.py tests/test_trainer_gpu.py tests/test_cfg5_gpu.py -x -q -m gpu -k "gelu or vit or bert or clip or lit or cfg5 or mlp or trainer" > $out/tests_gelu.txt 2>&1; echo "tests rc=$?"; tail -n 2 $out/tests_gelu.txt
for r in 1 2; do
  echo "== pre $r"; CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_pre.so timeout 400 python scripts/gemm_microbench.py --chunk 2048 --reps 10 2>&1 | grep "gelu (\|fc1 plain"
  echo "== new $r"; timeout 400 python scripts/gemm_microbench.py --chunk 2048 --reps 10 2>&1 | grep "gelu (\|fc1 plain"
done
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_pre.so timeout 900 python bench.py --steps 3 --warmup 1 --only-config-legs cfg1,lit,clip > $out/legs_gelu_pre_$r.log 2>&1
  timeout 900 python bench.py --steps 3 --warmup 1 --only-config-legs cfg1,lit,clip > $out/legs_gelu_new_$r.log 2>&1
done
python - <<'PY'
import json
for n in ('pre_1', 'new_1', 'pre_2', 'new_2'):
    d = json.loads([l for l in open(f'gpurun_out/r6m/legs_gelu_{n}.log') if l.startswith('{')][-1])
    print(n, "  ".join(f"{k} {d[k]['value']:.1f} ({d[k]['ms_per_step']:.1f} ms)" for k in ('cfg1', 'lit', 'clip') if k in d))
PY
