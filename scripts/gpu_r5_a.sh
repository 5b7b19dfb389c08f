#!/bin/bash
# Round 5, GPU call A: the host-side changes (accuracy out of the loss kernel, tracker, projection head, calibration, N > 1
# reporting) on the real kernels + same-box A/B of the SwiGLU-backward epilogue variants + the dQ read-modify-write probe.
set -u
mkdir -p gpurun_out/r5a
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r5a
(rocminfo | grep -m3 -E "Marketing|gfx950|Compute Unit"; rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -iE "power|sclk|mclk") > $O/host_info.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
timeout 60 python scripts/box_calibration.py > $O/box_calibration.json 2>$O/box_calibration.err; cat $O/box_calibration.json | cut -c1-600
timeout 400 python scripts/lib_ab.py --libs base,hi1,hi2,op2 --cases swiglu_bwd --rounds 9 > $O/ab_swiglu_bwd.txt 2>&1; cat $O/ab_swiglu_bwd.txt
timeout 300 python scripts/lib_ab.py --libs base --cases attn_fwd,attn_bwd,attn_bwd_ragged,qkv_fwd,out_fwd_res,fc2_fwd_res,swiglu_fwd_save --rounds 5 > $O/ab_base.txt 2>&1; cat $O/ab_base.txt
timeout 200 python scripts/dq_rmw_probe.py > $O/dq_rmw_probe.txt 2>&1; cat $O/dq_rmw_probe.txt
timeout 200 python scripts/attn_microbench.py --tokens 589824 --seqs 2048 > $O/attn_s2048.txt 2>&1; tail -2 $O/attn_s2048.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_short.log 2>&1; grep "^{" $O/bench_short.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in d if k.startswith('box_') or 'frac' in k or k in ('value','ms_per_step','schedule')})
print(d['box'])"
CX_TEST_TWO_TENANTS=1 timeout 400 python -m pytest tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape -x -q -s > $O/two_tenants.txt 2>&1; tail -6 $O/two_tenants.txt | cut -c1-1500
