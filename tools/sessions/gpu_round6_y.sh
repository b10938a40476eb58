#!/bin/bash
# round 6, session y: rows of 32768 words (workload F) -- the composed product's pair on incomplete transforms (build_row32k level 2):
# parity, then held rates level 0 / 2 alternated on the same box (F at the bench batch 1024 and at 4096); the new fused-row equality test
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_incomplete.py tests/test_gpu_fused.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/r06_F_incomplete_tests.txt
cat gpurun_out/r06_F_incomplete_tests.txt
{
for rep in 1 2 3; do
  for lv in 0 2; do
    echo -n "F batch 1024 level $lv: "; NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py 32768 2 1024 3
    echo -n "F batch 4096 level $lv: "; NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py 32768 2 4096 3
  done
done
} > gpurun_out/r06_F_incomplete_ab.txt 2>&1
cat gpurun_out/r06_F_incomplete_ab.txt
