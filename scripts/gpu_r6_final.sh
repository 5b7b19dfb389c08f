#!/bin/bash
# Round-6 evidence run (one gpurun call) of the code state at the time of the call: smoke, the GPU suite, the calibrated bench line with
# every leg, rocprofv3 kernel stats of the headline leg and of the cfg1 / lit / clip / cfg3 legs, PMC traffic + SQ counters, microbenchmarks
# (warm clocks), the phase trace of the fused S <= 128 attention backward, the plain GEMM against the vendor BLAS on the seven shapes, and the
# same-box A/B of the headline step and the config legs against the 32x32x16 GEMMs (variant m32).  Everything lands in gpurun_out/final6/;
# scripts/collect_r6.sh copies the summaries to profiles/r6_*.
set -u
mkdir -p gpurun_out/final6
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/final6
(rocminfo | grep -m3 -E "Marketing|gfx950|Compute Unit"; lscpu | grep -E "Model name|^CPU\(s\)"; rocm-smi --showmaxpower 2>/dev/null | grep -i power) > $O/host_info.txt 2>&1
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout 120 python scripts/box_calibration.py > $O/box_calibration.json 2>/dev/null; cut -c1-300 $O/box_calibration.json
# --- HBM traffic of the GEMM family (separate passes, guide's corrections), at the bench's launch sizes
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o b -- python $R/bench.py --steps 1 --warmup 0 --global-batch 4096 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $O/pmc_$ctr.log 2>&1)
  python scripts/pmc_summary.py $(find $O/pmc_$ctr -name "*counter_collection.csv" | head -1) $ctr > $O/pmc_${ctr}_summary.txt 2>&1
  rm -rf $O/pmc_$ctr
done
python scripts/pmc_traffic.py $O/pmc_FETCH_SIZE_summary.txt $O/pmc_WRITE_SIZE_summary.txt 4096 > $O/pmc_gemm_traffic.json 2>&1
python -c "import json; json.load(open('$O/pmc_gemm_traffic.json'))" && cp $O/pmc_gemm_traffic.json profiles/r6_pmc_gemm_traffic.json
# --- the bench line (reads the traffic file just written)
timeout 1800 python bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1; grep "^{" $O/bench.log | cut -c1-300
# --- kernel stats of the headline leg
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $O/prof.log 2>&1)
t=$(find $O/prof -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" > $O/kernel_summary.txt 2>&1
f=$(find $O/prof -name "*kernel_stats*.csv" | head -1); [[ -n "$f" ]] && cp "$f" $O/kernel_stats.csv
rm -rf $O/prof
head -12 $O/kernel_summary.txt | cut -c1-150
# --- SQ / MFMA counters over one step
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
(cd /tmp && timeout 600 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/pmc_sq -o g -- python $R/bench.py --steps 1 --warmup 0 --global-batch 4096 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $O/pmc_sq.log 2>&1)
python scripts/pmc_multi.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) > $O/pmc_sq_summary.txt 2>&1
rm -rf $O/pmc_sq
# --- kernel stats per config leg
for leg in cfg1 lit clip cfg3; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$leg -o p -- python $R/bench.py --steps 2 --only-config-legs $leg > $O/prof_$leg.log 2>&1)
  t=$(find $O/prof_$leg -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" 26 > $O/kernel_summary_$leg.txt 2>&1
  rm -rf $O/prof_$leg
done
# --- microbenchmarks (warm clocks: every shape runs >= 0.5 s before it is timed)
timeout 400 python scripts/gemm_microbench.py --chunk 2048 --reps 10 > $O/gemm_microbench.txt 2>&1; tail -10 $O/gemm_microbench.txt
timeout 300 python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $O/v6_vs_vendor_time.txt 2>&1; tail -9 $O/v6_vs_vendor_time.txt
{ echo "## shipped kernels, no rotation tables (what the engine passes beyond S = 128; image towers): S <= 128 single pass, S <= 256 K/V-resident forward, second-generation streaming kernels"; timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 128,197,256,512,2048,8192 --rotary 0;
  echo "## round 1's streaming kernels on the same box (cx_attn_set_fwd_long(0), cx_attn_set_bwd_long(0); 197 / 256: max_seqlen padded past 256 for the forward)"; timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 512,2048,8192 --rotary 0 --fwd-long 0 --bwd-long 0; timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197,256 --rotary 0 --fwd-long 0 --bwd-long 0 --max-seqlen-pad 100;
  echo "## with rotation tables (S = 128: the metric's kernels rotate on load; beyond 128 the API's rotate-on-load path keeps round 1's kernels -- the engine pre-rotates instead)"; timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 128,2048 --rotary 1; } > $O/attn_microbench.txt 2>&1
grep -v amdgpu.ids $O/attn_microbench.txt | tail -22
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 128,197,2048 --rotary 0 --pdrop 0.1 > $O/attn_microbench_dropout.txt 2>&1
# --- phase trace of the shipped fused S <= 128 attention backward
if [[ -f contrastors_amd/lib/variants/libcontrastors_hip_dev_attntrace.so ]]; then
  CONTRASTORS_HIP_DEV_LIB=contrastors_amd/lib/variants/libcontrastors_hip_dev_attntrace.so timeout 200 python scripts/attn_trace.py > $O/attn_bwd_s128_phase_trace.txt 2>&1
fi
# --- the headline step and the config legs with the 32x32x16 GEMMs of commit ee8a813 (variant m32, see scripts/gpu_r6_m16.sh), same box,
#     alternating libraries (host code identical).  (The A/B against round 5's attention / fusion routing -- variant r5routes -- was taken at
#     ee8a813 and is on file: profiles/r6_legs_ab_r6_vs_r5_routes.txt.)
L=contrastors_amd/lib
if [[ -f $L/variants/libcontrastors_hip_m32.so ]]; then
  i=0
  for v in base m32 base m32; do
    i=$((i+1))
    if [[ $v == base ]]; then unset CONTRASTORS_HIP_LIB; else export CONTRASTORS_HIP_LIB=$L/variants/libcontrastors_hip_$v.so; fi
    timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $O/step_ab_${v}_$i.log 2>&1
    timeout 900 python bench.py --steps 3 --warmup 1 --only-config-legs cfg1,lit,clip,cfg3 > $O/legs_ab_${v}_$i.log 2>&1
  done
  unset CONTRASTORS_HIP_LIB
fi
