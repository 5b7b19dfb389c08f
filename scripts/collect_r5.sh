#!/bin/bash
# copy the summaries of gpurun_out/final5 (scripts/gpu_r5_final.sh) into profiles/ under their round-5 names
set -u
O=gpurun_out/final5
cp $O/host_info.txt profiles/r5_host_info.txt
grep "^{" $O/bench.log | tail -1 > profiles/r5_bench_gb16384_n1.json
cp $O/kernel_summary.txt profiles/r5_kernel_summary_gb16384.txt
cp $O/kernel_stats.csv profiles/r5_rocprofv3_kernel_stats_gb16384.csv
cp $O/pmc_FETCH_SIZE_summary.txt profiles/r5_pmc_FETCH_SIZE_summary.txt
cp $O/pmc_WRITE_SIZE_summary.txt profiles/r5_pmc_WRITE_SIZE_summary.txt
cp $O/pmc_gemm_traffic.json profiles/r5_pmc_gemm_traffic.json
cp $O/pmc_sq_summary.txt profiles/r5_pmc_sq_step_summary.txt
for leg in cfg1 lit clip; do cp $O/kernel_summary_$leg.txt profiles/r5_kernel_summary_$leg.txt; done
cp $O/gemm_microbench.txt profiles/r5_microbench_gemm2048.txt
cp $O/attn_microbench.txt profiles/r5_microbench_attention.txt
cp $O/attn_microbench_dropout.txt profiles/r5_microbench_attention_dropout_final.txt
tail -3 $O/gpu_tests.txt > profiles/r5_gpu_tests.txt; tail -1 $O/two_tenants.txt >> profiles/r5_gpu_tests.txt; tail -1 $O/smoke.log >> profiles/r5_gpu_tests.txt
cp $O/box_calibration.json profiles/r5_box_calibration.json
cp $O/step_ab_r5_vs_r4_kernels.txt profiles/r5_step_ab_r5_vs_r4_kernels.txt
ls profiles | grep r5_
