#!/bin/bash
# Plain epilogue with its stores / residual loads spread over the next pass's convert work (built default) vs back to back (rs2)
set -u
out=gpurun_out/r6m; mkdir -p $out
V=contrastors_amd/lib/variants
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or wgrad or linear or residual" > $out/tests_gemm_g.txt 2>&1; echo "gemm tests rc=$?"; tail -n 1 $out/tests_gemm_g.txt
python scripts/gemm_ablate.py > $out/ablate_stilv.txt 2>&1
grep -v amdgpu.ids $out/ablate_stilv.txt | cut -c1-170 | grep "^==\|full kernel again\|without global\|no epilogue"
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_rs2.so python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_g_rs2_$r.txt 2>&1
  python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_g_stilv_$r.txt 2>&1
done
tail -n 9 $out/time_g_rs2_2.txt $out/time_g_stilv_2.txt
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_rs2.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $out/step_g_rs2_$r.json 2>/dev/null
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $out/step_g_stilv_$r.json 2>/dev/null
done
python - <<'PY'
import json
for n in ('rs2_1', 'stilv_1', 'rs2_2', 'stilv_2'):
    d = json.loads(open(f'gpurun_out/r6m/step_g_{n}.json').read().strip().splitlines()[-1])
    print(n, round(d['value'], 1), round(d['ms_per_step'], 1), d['roofline']['achieved'])
PY
