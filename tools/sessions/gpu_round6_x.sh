#!/bin/bash
# round 6, session x: same-box A/B of the generated wave-per-row fused kernels (32- and 64-bit limbs) against the compiled one-pass
# template (nflhip_debug_fused_grid(4) = tools/lwe_demo.py --grid 4), LWE demo, alternated twice
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
{
for rep in 1 2; do
  for g in 4 0; do
    for cfgs in "32 1024 2 65536" "32 1024 1 131072" "32 2048 2 32768" "32 4096 2 16384" "64 1024 2 32768" "64 2048 2 16384"; do
      set -- $cfgs
      PYTHONPATH=$here python tools/lwe_demo.py --limb-bits $1 --degree $2 --nmoduli $3 --batch $4 --plan fused --fixed-key --reps 10 --grid $g 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$( [ $g = 4 ] && echo template || echo generated ) lwe u$1 $2 $3 enc/s', d['encryptions_per_s'], 'dec/s', d['decryptions_per_s'], d['decrypts_to_zero'], d['digest'])"
    done
  done
done
} > gpurun_out/r06_lwe_rows_ab.txt 2>&1
cat gpurun_out/r06_lwe_rows_ab.txt
