#!/bin/bash
# round 5, session j: the wave-per-row fused forward kernels -- parity, then the LWE demo at the reference's short-row configurations
cd "$(dirname "$0")/../.."
here=$(pwd)
export PYTHONPATH=$here
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fused.py tests/test_gpu_samplers.py -m gpu -q 2>&1 | cut -c1-300 | tail -25 > gpurun_out/r05_j_tests.txt
tail -8 gpurun_out/r05_j_tests.txt
{
for cfg in "32 1024 2 65536" "64 1024 2 32768" "64 2048 2 16384" "32 4096 2 16384" "32 2048 2 32768"; do
  set -- $cfg
  for plan in unfused fused; do
    python tools/lwe_demo.py --limb-bits $1 --degree $2 --nmoduli $3 --batch $4 --plan $plan --fixed-key --reps 10 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['limb_bits'], d['degree'], d['nmoduli'], d['plan'], 'enc/s', d['encryptions_per_s'], 'dec/s', d['decryptions_per_s'], 'ok', d['decrypts_to_zero'], d['digest'])"
  done
done
} > gpurun_out/r05_lwe_short_rows.txt 2>&1
cat gpurun_out/r05_lwe_short_rows.txt
