#!/bin/bash
# 16x16x32 main loop: ablation incl. "DMA from wave 0 only", then the headline step with the 32x32x16 library of ee8a813 and the product, alternating.
set -u
out=gpurun_out/r6m; mkdir -p $out
python scripts/gemm_ablate.py > $out/ablate_m16b.txt 2>&1; echo "ablate rc=$?"; grep -v amdgpu.ids $out/ablate_m16b.txt | cut -c1-170 | head -20
V=contrastors_amd/lib/variants
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
