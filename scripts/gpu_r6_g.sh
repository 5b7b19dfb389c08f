#!/bin/bash
set -u
mkdir -p gpurun_out/r6g
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
O=gpurun_out/r6g
echo "## S = 197 / 256 forward: streaming long kernel (cx_attn_set_fwd_s128(0)) vs the K/V-resident single pass" > $O/attn_fwd_197.txt
timeout 200 python scripts/attn_microbench.py --tokens 262144 --seqs 197,256 --rotary 0 --fwd-mode 0 >> $O/attn_fwd_197.txt 2>&1
timeout 200 python scripts/attn_microbench.py --tokens 262144 --seqs 197,256 --rotary 0 >> $O/attn_fwd_197.txt 2>&1
grep -v amdgpu $O/attn_fwd_197.txt
timeout 600 python bench.py --steps 3 --warmup 1 --only-config-legs clip,lit > $O/legs.log 2>&1
grep "^{" $O/legs.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('lit','clip'):
    r=d.get(k,{})
    print(k, round(r.get('value',0),1), round(r.get('ms_per_step',0),1),'ms', (r.get('roofline') or {}).get('frac'), (r.get('selective_checkpointing') or {}).get('value'))
"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/prof_clip -o p -- python $OLDPWD/bench.py --steps 2 --only-config-legs clip > /dev/null 2>&1)
