#!/bin/bash
# Round 5, GPU call K: the streaming dK/dV kernel's dropout form (prefetch + 2 workgroups per CU, shipped) against its round-4 form, with and without rotation
# tables; the plain form is round 4's again (sanity: equal to pf0).
set -u
mkdir -p gpurun_out/r5k
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5k
timeout 300 python scripts/lib_ab.py --libs base,pf0 --cases attn_bwd --seq 2048 --rotary 0 --chunk 1024 --rounds 5 --reps 3 > $O/ab_plain_sanity.txt 2>&1; grep -v "^/opt" $O/ab_plain_sanity.txt
for rot in 0 1; do
  for lib in shipped pf0; do
    if [[ $lib == pf0 ]]; then export CONTRASTORS_HIP_DEV_LIB=$PWD/contrastors_amd/lib/variants/libcontrastors_hip_dev_pf0.so; else unset CONTRASTORS_HIP_DEV_LIB; fi
    echo "== dropout 0.1, rotary $rot, $lib" | tee -a $O/dropout_dkv_ab.txt
    timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197,512,2048 --pdrop 0.1 --rotary $rot 2>&1 | grep -v "^/opt" | tail -3 | tee -a $O/dropout_dkv_ab.txt
  done
done
