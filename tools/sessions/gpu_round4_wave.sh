#!/bin/bash
# round 4: INTT(b -+ a k) on the wave-per-row kernels (rows of 1024 / 2048 words, 4096 for 32-bit limbs): parity, the demo at the reference's 32-bit configuration
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4wave
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -3
for cfg in "32 1024 2 65536" "64 1024 2 65536" "64 2048 2 16384" "32 4096 2 8192"; do
  set -- $cfg
  for plan in unfused fused; do
    timeout 300 python tools/lwe_demo.py --limb-bits $1 --degree $2 --nmoduli $3 --batch $4 --plan $plan --reps 10 --fixed-key 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['limb_bits'], d['degree'], d['nmoduli'], d['plan'], d['encryptions_per_s'], d['decryptions_per_s'], d['decrypts_to_zero'], d['digest']['dec'])"
  done
done | tee $O/lwe_wave.txt
