#!/bin/bash
# copy the summaries of gpurun_out/final6 (scripts/gpu_r6_final.sh) into profiles/ under their round-6 names
set -u
O=gpurun_out/final6
cp $O/host_info.txt profiles/r6_host_info.txt
grep "^{" $O/bench.log | tail -1 > profiles/r6_bench_gb16384_n1.json
cp $O/kernel_summary.txt profiles/r6_kernel_summary_gb16384.txt
cp $O/kernel_stats.csv profiles/r6_rocprofv3_kernel_stats_gb16384.csv
cp $O/pmc_FETCH_SIZE_summary.txt profiles/r6_pmc_FETCH_SIZE_summary.txt
cp $O/pmc_WRITE_SIZE_summary.txt profiles/r6_pmc_WRITE_SIZE_summary.txt
cp $O/pmc_gemm_traffic.json profiles/r6_pmc_gemm_traffic.json
cp $O/pmc_sq_summary.txt profiles/r6_pmc_sq_step_summary.txt
for leg in cfg1 lit clip cfg3; do cp $O/kernel_summary_$leg.txt profiles/r6_kernel_summary_$leg.txt; done
grep -v amdgpu.ids $O/gemm_microbench.txt > profiles/r6_microbench_gemm2048.txt
grep -v amdgpu.ids $O/attn_microbench.txt > profiles/r6_microbench_attention.txt
grep -v amdgpu.ids $O/attn_microbench_dropout.txt > profiles/r6_microbench_attention_dropout.txt
grep -v amdgpu.ids $O/attn_bwd_s128_phase_trace.txt > profiles/r6_attn_bwd_s128_phase_trace.txt
tail -3 $O/gpu_tests.txt > profiles/r6_gpu_tests.txt; tail -1 $O/smoke.log >> profiles/r6_gpu_tests.txt
cp $O/box_calibration.json profiles/r6_box_calibration.json
grep -v amdgpu.ids $O/v6_vs_vendor_time.txt > profiles/r6_v6_vs_vendor_time.txt
{ echo "# Same-box A/B at the round's last code state: the product library (16x16x32 main loop, LDS-DMA / fragment reads spread over the K-tile) against"
  echo "# variant m32 = the library of commit ee8a813 (32x32x16 main loop, requests back to back); host code identical, alternating runs"
  echo "# (scripts/gpu_r6_final.sh): bench.py --steps 3 --warmup 1 on the headline step, then --only-config-legs cfg1,lit,clip,cfg3."
  for v in base m32; do for f in $O/step_ab_${v}_*.log; do [[ -f $f ]] && python3 - "$v" "$f" <<'PY'
import json,sys
v,f=sys.argv[1],sys.argv[2]
d=json.loads([l for l in open(f) if l.startswith('{')][-1])
print(f"{v:5s} headline {d['value']:.1f} pairs/s, {d['ms_per_step']:.1f} ms/step, GEMM family {d['roofline']['achieved']:.1f} TFLOP/s")
PY
  done; done
  for v in base m32; do for f in $O/legs_ab_${v}_*.log; do [[ -f $f ]] && python3 - "$v" "$f" <<'PY'
import json,sys
v,f=sys.argv[1],sys.argv[2]
d=json.loads([l for l in open(f) if l.startswith('{')][-1])
def sel(k):
    s=(d[k].get('selective_checkpointing') or {}).get('value')
    return f" (selective checkpointing {s:.1f})" if s else ""
print(f"{v:5s}", "   ".join(f"{k} {d[k]['value']:.1f} {d[k]['unit'].split()[0]}/s, {d[k]['ms_per_step']:.1f} ms{sel(k)}" for k in ("cfg1","cfg3","lit","clip") if k in d))
PY
  done; done; } > profiles/r6_step_legs_ab_m16_vs_m32.txt
ls profiles | grep r6_
