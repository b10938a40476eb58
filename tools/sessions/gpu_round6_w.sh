#!/bin/bash
# round 6, session w: the generated transform-fused wave-per-row kernels for 32-bit limbs (rows of 1024 / 2048 / 4096 words;
# tools/gen_row1024_u32_asm.py build_fwd_fma / build_fma_inv): parity (fused + fuzz + row tests), the LWE demo on those rings against the
# compiled template (NFLHIP_VARIANT=hipcc), then the whole gpu suite
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fuzz.py tests/test_gpu_u32_asm.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/r06_rows_u32_fused_tests.txt
cat gpurun_out/r06_rows_u32_fused_tests.txt
{
for rep in 1 2; do
  for v in hipcc asm; do
    for cfgs in "1024 2 65536" "1024 1 131072" "2048 2 32768" "4096 2 16384"; do
      set -- $cfgs
      if [ $v = hipcc ]; then export NFLHIP_VARIANT=hipcc; else unset NFLHIP_VARIANT; fi
      PYTHONPATH=$here python tools/lwe_demo.py --limb-bits 32 --degree $1 --nmoduli $2 --batch $3 --plan fused --fixed-key --reps 10 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v lwe u32', d['degree'], d.get('nmoduli', $2), 'enc/s', d['encryptions_per_s'], 'dec/s', d['decryptions_per_s'], d['decrypts_to_zero'], d['digest'])"
    done
  done
done
unset NFLHIP_VARIANT
} > gpurun_out/r06_lwe_rows_u32.txt 2>&1
cat gpurun_out/r06_lwe_rows_u32.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06_gputests_w.txt
cat gpurun_out/r06_gputests_w.txt
