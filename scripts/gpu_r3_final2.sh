#!/bin/bash
# Round-3 evidence run of the FINAL code state (selective checkpointing, gate-only save of the gated MLP): one gpurun call.
# Full GPU tests + smoke, PMC traffic passes, the bench line, rocprofv3 kernel stats of the headline leg, SQ / MFMA counters,
# kernel stats of the cfg 3 leg with selective checkpointing, GEMM microbenchmark.  The legs whose kernels did not change
# since scripts/gpu_r3_final.sh (lit, clip PMC traffic, attention / ViT / long-sequence microbenchmarks) keep their files.
# Everything lands in gpurun_out/final3b/; scripts/collect_r3b.sh copies the summaries to profiles/r3_*.
set -u
mkdir -p gpurun_out/final3b
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/final3b
(rocminfo | grep -m3 -E "Marketing|gfx950|Compute Unit"; lscpu | grep -E "Model name|^CPU\(s\)"; rocm-smi --showmaxpower 2>/dev/null | grep -i power) > $O/host_info.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cp gpurun_out/kernel_report.jsonl $O/kernel_report.jsonl 2>/dev/null
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o b -- python $R/bench.py --steps 1 --warmup 0 --global-batch 2048 --no-cpu-baseline --no-extra-legs > $O/pmc_$ctr.log 2>&1)
  python scripts/pmc_summary.py $(find $O/pmc_$ctr -name "*counter_collection.csv" | head -1) $ctr > $O/pmc_${ctr}_summary.txt 2>&1
  rm -rf $O/pmc_$ctr
done
python scripts/pmc_traffic.py $O/pmc_FETCH_SIZE_summary.txt $O/pmc_WRITE_SIZE_summary.txt 2048 > $O/pmc_gemm_traffic.json 2>&1
python -c "import json; json.load(open('$O/pmc_gemm_traffic.json'))" && cp $O/pmc_gemm_traffic.json profiles/r3_pmc_gemm_traffic.json
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1; grep "^{" $O/bench.log | cut -c1-300
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/prof.log 2>&1)
t=$(find $O/prof -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" > $O/kernel_summary.txt 2>&1
f=$(find $O/prof -name "*kernel_stats*.csv" | head -1); [[ -n "$f" ]] && cp "$f" $O/kernel_stats.csv
rm -rf $O/prof
head -12 $O/kernel_summary.txt | cut -c1-150
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
(cd /tmp && timeout 400 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/pmc_sq -o g -- python $R/bench.py --steps 1 --warmup 0 --global-batch 2048 --no-cpu-baseline --no-extra-legs > $O/pmc_sq.log 2>&1)
python scripts/pmc_multi.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) > $O/pmc_sq_summary.txt 2>&1
rm -rf $O/pmc_sq
# cfg 3 with selective checkpointing: 1 literal warm-up step (the HBM peak is measured) + 3 steps with every block kept
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3s -o p -- python $R/bench.py --steps 3 --only-config-legs cfg3_selective > $O/prof_cfg3s.log 2>&1)
t=$(find $O/prof_cfg3s -name "*kernel_trace*.csv" | head -1); [[ -n "$t" ]] && python scripts/prof_summary.py "$t" 24 > $O/kernel_summary_cfg3_selective.txt 2>&1
rm -rf $O/prof_cfg3s
timeout 300 python scripts/gemm_microbench.py --chunk 2048 --reps 6 > $O/gemm_microbench.txt 2>&1; tail -9 $O/gemm_microbench.txt
