#!/bin/bash
# Diagnostic GPU call (round 2): microbench baseline, stream-concurrency probe, PMC passes over the GEMM microbench.
set -u
mkdir -p gpurun_out/diag
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/diag
timeout 300 python scripts/gemm_microbench.py --chunk 1024 --reps 10 > $O/microbench_gemm1024.txt 2>&1
tail -13 $O/microbench_gemm1024.txt
timeout 300 python scripts/concurrency_probe.py > $O/concurrency.txt 2>&1
cat $O/concurrency.txt | tail -6
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc$i -o g -- python $R/scripts/gemm_microbench.py --chunk 1024 --reps 2 > $O/pmc$i.log 2>&1; echo "pmc$i exit $?" >> $O/pmc$i.log)
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  [[ -n "$f" ]] && python scripts/pmc_multi.py "$f" gemm > $O/pmc${i}_summary.txt 2>&1
  rm -rf $O/pmc$i
  tail -2 $O/pmc$i.log
done
head -40 $O/pmc1_summary.txt
