#!/bin/bash
# copy the summaries of gpurun_out/final3b (scripts/gpu_r3_final2.sh) into profiles/ under their round-3 names
set -u
O=gpurun_out/final3b
cp $O/host_info.txt profiles/r3_host_info.txt
grep "^{" $O/bench.log | tail -1 > profiles/r3_bench_gb16384_n1.json
cp $O/kernel_summary.txt profiles/r3_kernel_summary_gb16384.txt
cp $O/kernel_stats.csv profiles/r3_rocprofv3_kernel_stats_gb16384.csv
cp $O/pmc_FETCH_SIZE_summary.txt profiles/r3_pmc_FETCH_SIZE_summary.txt
cp $O/pmc_WRITE_SIZE_summary.txt profiles/r3_pmc_WRITE_SIZE_summary.txt
cp $O/pmc_gemm_traffic.json profiles/r3_pmc_gemm_traffic.json
cp $O/pmc_sq_summary.txt profiles/r3_pmc_sq_step_summary.txt
cp $O/kernel_summary_cfg3_selective.txt profiles/r3_kernel_summary_cfg3_selective.txt
cp $O/gemm_microbench.txt profiles/r3_microbench_gemm2048.txt
tail -3 $O/pytest.log > profiles/r3_gpu_tests.txt; tail -1 $O/smoke.log >> profiles/r3_gpu_tests.txt
ls profiles | grep r3_
