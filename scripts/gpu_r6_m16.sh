#!/bin/bash
# Round 6, the 16x16x32 / spread-schedule GEMMs against the 32x32x16 kernels they replace, one box, one call:
#   contrastors_amd/lib/variants/libcontrastors_hip_m32.so = the product library built from commit ee8a813 (git worktree add /tmp/w ee8a813;
#   python -m contrastors_amd.build there; copy lib/libcontrastors_hip.so) -- a build product, not tracked.
# GEMM parity tests, scripts/v6_vs_vendor.py time with both libraries (alternating processes), the headline step with both.
# -> profiles/r6_gemm_mfma16_ab.txt, r6_gemm_v6_m16_schedule_ab.txt (the in-situ experiment that started it: r6_gemm_mfma16_in_situ.txt)
set -u
out=gpurun_out/r6m; mkdir -p $out
V=contrastors_amd/lib/variants
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or wgrad or swiglu or gelu or act_bwd or linear" > $out/tests_gemm.txt 2>&1
echo "gemm tests rc=$?"; tail -n 2 $out/tests_gemm.txt
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_m32.so python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_m32_$r.txt 2>&1
  python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_m16_$r.txt 2>&1
done
tail -n 10 $out/time_m32_2.txt $out/time_m16_2.txt
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_m32.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $out/step_m32_$r.json 2>/dev/null
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $out/step_m16_$r.json 2>/dev/null
done
python - <<'PY'
import json
for n in ('m32_1', 'm16_1', 'm32_2', 'm16_2'):
    d = json.loads(open(f'gpurun_out/r6m/step_{n}.json').read().strip().splitlines()[-1])
    print(n, round(d['value'], 1), round(d['ms_per_step'], 1), d['roofline']['achieved'])
PY
