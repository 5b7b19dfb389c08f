#!/bin/bash
# Per-wave DMA windows (CX_MID): parity, ablation cycles, then product vs the un-staggered 16x16x32 library (m16a) vs the 32x32x16 one (m32), alternating.
set -u
out=gpurun_out/r6m; mkdir -p $out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or wgrad or swiglu or gelu or act_bwd or linear" > $out/tests_gemm_d.txt 2>&1
echo "gemm tests rc=$?"; tail -n 2 $out/tests_gemm_d.txt
python scripts/gemm_ablate.py > $out/ablate_stag.txt 2>&1; grep -v amdgpu.ids $out/ablate_stag.txt | cut -c1-170 | grep "==\|full kernel again\|no epilogue\|no DMA  \|MFMA only, no"
V=contrastors_amd/lib/variants
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_m16a.so python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_m16a_$r.txt 2>&1
  python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $out/time_stag_$r.txt 2>&1
done
tail -n 9 $out/time_m16a_2.txt $out/time_stag_2.txt
for r in 1 2; do
  CONTRASTORS_HIP_LIB=$V/libcontrastors_hip_m16a.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $out/step_m16a_$r.json 2>/dev/null
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $out/step_stag_$r.json 2>/dev/null
done
python - <<'PY'
import json
for n in ('m16a_1', 'stag_1', 'm16a_2', 'stag_2'):
    d = json.loads(open(f'gpurun_out/r6m/step_{n}.json').read().strip().splitlines()[-1])
    print(n, round(d['value'], 1), round(d['ms_per_step'], 1), d['roofline']['achieved'])
PY
