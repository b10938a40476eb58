#!/bin/bash
# round 4: the LWE demo at the reference's OWN test configurations (tests/CMakeLists.txt:1-7: 1024 / 60 bits / uint32_t, 8192 / 124 / uint64_t,
# 32768 / 124 / uint64_t), operator by operator and through the fused entries (one generated kernel at 8192; composed from the plain kernels elsewhere)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4ref
mkdir -p $O
for cfg in "32 1024 2 65536" "64 8192 2 8192" "64 32768 2 1024"; do
  set -- $cfg
  for plan in unfused fused; do
    timeout 300 python tools/lwe_demo.py --limb-bits $1 --degree $2 --nmoduli $3 --batch $4 --plan $plan --reps 10 --fixed-key 2>/dev/null >> $O/lwe_reference_configs.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r4ref/lwe_reference_configs.jsonl'):
    d = json.loads(l); print(d['limb_bits'], d['degree'], d['nmoduli'], d['batch'], d['plan'], d['encryptions_per_s'], d['decryptions_per_s'], d['decrypts_to_zero'], d['digest']['dec'])
PY
