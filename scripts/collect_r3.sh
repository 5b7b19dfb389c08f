#!/bin/bash
# copy the summaries of gpurun_out/final3 (scripts/gpu_r3_final.sh) into profiles/ under their round-3 names
set -u
O=gpurun_out/final3
cp $O/host_info.txt profiles/r3_host_info.txt
grep "^{" $O/bench.log | tail -1 > profiles/r3_bench_gb16384_n1.json
cp $O/kernel_summary.txt profiles/r3_kernel_summary_gb16384.txt
cp $O/kernel_stats.csv profiles/r3_rocprofv3_kernel_stats_gb16384.csv
cp $O/pmc_FETCH_SIZE_summary.txt profiles/r3_pmc_FETCH_SIZE_summary.txt
cp $O/pmc_WRITE_SIZE_summary.txt profiles/r3_pmc_WRITE_SIZE_summary.txt
cp $O/pmc_gemm_traffic.json profiles/r3_pmc_gemm_traffic.json
cp $O/pmc_sq_summary.txt profiles/r3_pmc_sq_step_summary.txt
for leg in cfg3 lit clip; do cp $O/kernel_summary_$leg.txt profiles/r3_kernel_summary_$leg.txt; done
cp $O/pmc_clip_FETCH_SIZE_summary.txt profiles/r3_pmc_clip_FETCH_SIZE_summary.txt
cp $O/pmc_clip_WRITE_SIZE_summary.txt profiles/r3_pmc_clip_WRITE_SIZE_summary.txt
cp $O/gemm_microbench.txt profiles/r3_microbench_gemm2048.txt
cp $O/attn_microbench.txt profiles/r3_microbench_attention.txt
cp $O/vit_microbench.txt profiles/r3_microbench_vit_b16.txt
cp $O/longseq_bench.txt profiles/r3_longseq_bench.txt
tail -3 $O/pytest.log > profiles/r3_gpu_tests.txt; tail -1 $O/smoke.log >> profiles/r3_gpu_tests.txt
ls profiles | grep r3_
