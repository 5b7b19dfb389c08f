#!/bin/bash
# SQ / LDS counters of the two NT GEMM structures on the same two launches (separate passes, --kernel-trace only)
cd "${GRAFT_REPO_ROOT:-.}"
R=$(pwd); O=$R/gpurun_out/r4; mkdir -p $O
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"
P3="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_v7_$i -o g -- python $R/scripts/gemm_v7_pmc_driver.py 1 > $O/pmc_v7_$i.log 2>&1)
  python scripts/pmc_multi.py $(find $O/pmc_v7_$i -name "*counter_collection.csv" | head -1) gemm_bf16_v > $O/pmc_v7_pass$i.txt 2>&1
  rm -rf $O/pmc_v7_$i
done
cat $O/pmc_v7_pass1.txt $O/pmc_v7_pass2.txt $O/pmc_v7_pass3.txt | grep -v "^   SQ_BUSY\|XCD)  " 
