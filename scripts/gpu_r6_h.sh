#!/bin/bash
set -u
mkdir -p gpurun_out/r6h
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6h
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 600 python -m pytest tests/test_dropout_gpu.py tests/test_vit_gpu.py tests/test_cfg3_gpu.py -q > $O/tests2.txt 2>&1; tail -2 $O/tests2.txt
timeout 300 python scripts/attn_microbench.py --tokens 262144 --seqs 197,256,300,2048 --rotary 0 > $O/attn.txt 2>&1; grep -v amdgpu $O/attn.txt
echo "## headline with GradCache chunk 4096 vs 2048 (same box)" > $O/chunk.txt
for c in 2048 4096 2048 4096; do
  CX_BENCH_CHUNK=$c timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-config-legs --no-calibration > $O/bench_$c.log 2>&1
  echo "chunk $c: $(grep '^{' $O/bench_$c.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "pairs/s", round(d["roofline"]["achieved"],1), "TF", round(d.get("peak_hbm_gb",0),1), "GB")')" | tee -a $O/chunk.txt
done
