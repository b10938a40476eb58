#!/bin/bash
# round 4: the inverse pipelines of 32768-word rows (nflhip_fused_{fms,fma}_inv32768_asm): parity, then the demo at the reference's largest configuration
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4f32k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for plan in unfused fused; do
  timeout 300 python tools/lwe_demo.py --degree 32768 --nmoduli 2 --batch 1024 --plan $plan --reps 10 --fixed-key 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['plan'], d['encryptions_per_s'], d['decryptions_per_s'], d['decrypts_to_zero'], d['digest']['dec'])"
done; done | tee $O/lwe_32768.txt
