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
{ echo "# Same-box A/B of the config legs: the product library against variant r5routes (round 5's kernel routing: round 1's streaming attention"
  echo "# kernels beyond S = 128 incl. the delta pass, no K/V-resident S <= 256 forward, standalone GELU backward + bias colsum; host code identical),"
  echo "# alternating runs of bench.py --only-config-legs cfg1,lit,clip,cfg3 --steps 3 (scripts/gpu_r6_final.sh)."
  for v in base r5routes; do for f in $O/legs_ab_${v}*.log; do [[ -f $f ]] && python3 - "$v" "$f" <<'PY'
import json,sys
v,f=sys.argv[1],sys.argv[2]
d=json.loads([l for l in open(f) if l.startswith('{')][-1])
def sel(k):
    s=(d[k].get('selective_checkpointing') or {}).get('value')
    return f" (selective checkpointing {s:.1f})" if s else ""
print(f"{v:9s}", "   ".join(f"{k} {d[k]['value']:.1f} {d[k]['unit'].split()[0]}/s, {d[k]['ms_per_step']:.1f} ms{sel(k)}" for k in ("cfg1","cfg3","lit","clip") if k in d))
PY
  done; done; } > profiles/r6_legs_ab_r6_vs_r5_routes.txt
ls profiles | grep r6_
