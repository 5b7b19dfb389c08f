#!/bin/bash
# Round-6 call A: like-for-like v6 vs vendor BLAS on the seven dense shapes (timing windows + three rocprofv3 counter passes).
set -u
mkdir -p gpurun_out/r6a
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r6a
timeout 120 python scripts/box_calibration.py > $O/box_calibration.json 2>/dev/null; cut -c1-600 $O/box_calibration.json
timeout 400 python scripts/v6_vs_vendor.py time --seconds 1.0 --rounds 2 > $O/v6_vs_vendor_time.txt 2>&1; tail -12 $O/v6_vs_vendor_time.txt
P1="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY FETCH_SIZE GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES WRITE_SIZE"
P3="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && timeout 500 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc$i -o w -- python $R/scripts/v6_vs_vendor.py workload --reps 3 > $O/pmc$i.log 2>&1)
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  [[ -n "$f" ]] && cp "$f" $O/pmc$i.csv
  rm -rf $O/pmc$i
  tail -2 $O/pmc$i.log
done
python scripts/v6_vs_vendor.py parse $O/pmc1.csv $O/pmc2.csv $O/pmc3.csv --reps 3 > $O/v6_vs_vendor_counters.txt 2>&1
head -60 $O/v6_vs_vendor_counters.txt
gzip -f $O/pmc1.csv $O/pmc2.csv $O/pmc3.csv
