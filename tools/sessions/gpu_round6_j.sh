#!/bin/bash
# round 6, session j: 32-bit limbs, the product on incomplete transforms (gen_row1024_u32_asm.py base_mul): parity, then level 0 / 2
# alternated on the same box at u32/1024/1 (BASELINE configs[0]'s shape), u32/1024/2, u32/2048/2, u32/4096/2
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_incomplete.py tests/test_gpu_u32_asm.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06_u32_incomplete_tests.txt
cat gpurun_out/r06_u32_incomplete_tests.txt
{
for rep in 1 2; do
  for cfgs in "1024 1 524288" "1024 2 262144" "2048 2 131072" "4096 2 65536"; do
    set -- $cfgs
    for lv in 0 2; do
      echo -n "u32 n $1 nm $2 level $lv: "; NFL_LIMB_BITS=32 NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py $1 $2 $3 2 2>/dev/null
    done
  done
done
} > gpurun_out/r06_u32_incomplete_ab.txt 2>&1
cat gpurun_out/r06_u32_incomplete_ab.txt
