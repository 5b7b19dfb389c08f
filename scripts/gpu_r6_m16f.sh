#!/bin/bash
# Read spread 2 + DMA spread 4 (built default) vs read spread 1 + DMA spread 4 (sp4, the previous build): parity, ablation cycles, timing
set -u
out=gpurun_out/r6m; mkdir -p $out
V=contrastors_amd/lib/variants
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or wgrad" > $out/tests_gemm_f.txt 2>&1; echo "gemm tests rc=$?"; tail -n 1 $out/tests_gemm_f.txt
python scripts/gemm_ablate.py > $out/ablate_rs2.txt 2>&1
for f in rs2; do echo "== $f"; grep -v amdgpu.ids $out/ablate_$f.txt | cut -c1-170 | grep "^==\|full kernel again\|no epilogue\|no DMA  \|no fragment\|MFMA only, no"; done
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_sp4.so python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_f_sp4_$r.txt 2>&1
  python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_f_rs2_$r.txt 2>&1
done
tail -n 9 $out/time_f_sp4_2.txt $out/time_f_rs2_2.txt
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_m32.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $out/step_f_m32_$r.json 2>/dev/null
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_sp4.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $out/step_f_sp4_$r.json 2>/dev/null
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $out/step_f_rs2_$r.json 2>/dev/null
done
python - <<'PY'
import json
for n in ('m32_1', 'sp4_1', 'rs2_1', 'm32_2', 'sp4_2', 'rs2_2'):
    d = json.loads(open(f'gpurun_out/r6m/step_f_{n}.json').read().strip().splitlines()[-1])
    print(n, round(d['value'], 1), round(d['ms_per_step'], 1), d['roofline']['achieved'])
PY
