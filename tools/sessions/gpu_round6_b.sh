#!/bin/bash
# round 6, session b: the metric product on incomplete transforms (tools/asmgen/incomplete.py) -- parity through the C ABI, then the
# same-box A/B of level 0 (complete, shipped) / 1 / 2 (nflhip_debug_polymul_level), alternated three times, 3 s per run
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_incomplete.py -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > gpurun_out/r06_incomplete_tests.txt
cat gpurun_out/r06_incomplete_tests.txt
{
for rep in 1 2 3; do
  for lv in 0 1 2; do
    python tools/ab_probe.py nfllib_amd/libnflhip.so 3 $lv 2>&1 | tail -1
  done
done
} > gpurun_out/r06_incomplete_ab.txt 2>&1
cat gpurun_out/r06_incomplete_ab.txt
