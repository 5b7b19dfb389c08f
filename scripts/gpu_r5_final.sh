#!/bin/bash
# Round-5 evidence run (one gpurun call) of the final code state: smoke, PMC traffic passes, the bench line with every leg,
# rocprofv3 kernel stats of the headline leg and of the cfg1 / lit / clip legs (the ones the v7 routing and the single-pass
# dropout attention changed), SQ / MFMA counters, microbenchmarks.  The GPU test run comes first (gpurun_out/final5/gpu_tests.txt).  Everything lands in gpurun_out/final5/; scripts/collect_r5.sh copies the summaries to profiles/r5_*.
set -u
mkdir -p gpurun_out/final5
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/final5
(rocminfo | grep -m3 -E "Marketing|gfx950|Compute Unit"; lscpu | grep -E "Model name|^CPU\(s\)"; rocm-smi --showmaxpower 2>/dev/null | grep -i power) > $O/host_info.txt 2>&1
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# --- the whole GPU suite (the two-tenant planner test on its own: it fills the device)
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
CX_TEST_TWO_TENANTS=1 timeout 400 python -m pytest tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape -q -s > $O/two_tenants.txt 2>&1; tail -1 $O/two_tenants.txt
timeout 60 python scripts/box_calibration.py > $O/box_calibration.json 2>/dev/null; cut -c1-400 $O/box_calibration.json
cp gpurun_out/kernel_report.jsonl $O/kernel_report.jsonl 2>/dev/null
# --- HBM traffic of the GEMM family (separate passes, guide's corrections), at the bench's launch sizes
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o b -- python $R/bench.py --steps 1 --warmup 0 --global-batch 2048 --no-cpu-baseline --no-extra-legs --no-calibration > $O/pmc_$ctr.log 2>&1)
  python scripts/pmc_summary.py $(find $O/pmc_$ctr -name "*counter_collection.csv" | head -1) $ctr > $O/pmc_${ctr}_summary.txt 2>&1
  rm -rf $O/pmc_$ctr
done
python scripts/pmc_traffic.py $O/pmc_FETCH_SIZE_summary.txt $O/pmc_WRITE_SIZE_summary.txt 2048 > $O/pmc_gemm_traffic.json 2>&1
python -c "import json; json.load(open('$O/pmc_gemm_traffic.json'))" && cp $O/pmc_gemm_traffic.json profiles/r5_pmc_gemm_traffic.json
# --- the bench line (reads the traffic file just written)
timeout 1500 python bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1; grep "^{" $O/bench.log | cut -c1-300
# --- kernel stats of the headline leg
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --no-calibration > $O/prof.log 2>&1)
t=$(find $O/prof -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" > $O/kernel_summary.txt 2>&1
f=$(find $O/prof -name "*kernel_stats*.csv" | head -1); [[ -n "$f" ]] && cp "$f" $O/kernel_stats.csv
rm -rf $O/prof
head -12 $O/kernel_summary.txt | cut -c1-150
# --- SQ / MFMA counters over one step
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
(cd /tmp && timeout 600 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/pmc_sq -o g -- python $R/bench.py --steps 1 --warmup 0 --global-batch 2048 --no-cpu-baseline --no-extra-legs --no-calibration > $O/pmc_sq.log 2>&1)
python scripts/pmc_multi.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) > $O/pmc_sq_summary.txt 2>&1
rm -rf $O/pmc_sq
# --- the legs this round changed: kernel stats per leg (dropout recipe of BERT-base in cfg1 / lit)
for leg in cfg1 lit clip; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$leg -o p -- python $R/bench.py --steps 2 --only-config-legs $leg > $O/prof_$leg.log 2>&1)
  t=$(find $O/prof_$leg -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" 24 > $O/kernel_summary_$leg.txt 2>&1
  rm -rf $O/prof_$leg
done
# --- microbenchmarks
timeout 300 python scripts/gemm_microbench.py --chunk 2048 --reps 6 > $O/gemm_microbench.txt 2>&1; tail -6 $O/gemm_microbench.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 > $O/attn_microbench.txt 2>&1; tail -5 $O/attn_microbench.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --pdrop 0.1 > $O/attn_microbench_dropout.txt 2>&1; tail -5 $O/attn_microbench_dropout.txt
# --- round 5's kernel changes against round 4's kernels, same box, alternating libraries (host code identical): variant r4k =
#     gemm_bf16_v6.hip with CX_V6_HI_EARLY=0 CX_V6_OPAQUE=0 + attention.hip with CX_ATTN_ROT_PRE=0 CX_ATTN_CS_FIRST=0 (scripts/build_variant.py)
L=contrastors_amd/lib
if [[ -f $L/variants/libcontrastors_hip_r4k.so ]]; then
  cp $L/libcontrastors_hip.so /tmp/base.so
  for v in base r4k base r4k; do
    if [[ $v == base ]]; then cp /tmp/base.so $L/libcontrastors_hip.so; else cp $L/variants/libcontrastors_hip_$v.so $L/libcontrastors_hip.so; fi
    timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --no-calibration > $O/step_ab_$v.log 2>&1
    echo "$v: $(grep '^{' $O/step_ab_$v.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("pairs/s", round(d["value"],1), "GEMM TF", round(d["roofline"]["achieved"],1), "median step ms", round(d["step_ms"]["median"],1))')" | tee -a $O/step_ab_r5_vs_r4_kernels.txt
  done
  cp /tmp/base.so $L/libcontrastors_hip.so
fi
