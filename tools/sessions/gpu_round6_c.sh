#!/bin/bash
# round 6, session c: long-row plans with their block products on incomplete transforms (E: pipeline / one-launch; F-sized xcd) --
# parity, then held rates at level 0 / 2 alternated on the same box; the full gpu suite with level 2 as the library's default
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_incomplete.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06_incomplete_tests_c.txt
cat gpurun_out/r06_incomplete_tests_c.txt
{
for rep in 1 2; do
  for lv in 0 2; do
    echo -n "E pipeline batch 128 level $lv: "; NFLHIP_XCD=0 NFL_POLYMUL_LEVEL=$lv timeout 200 python tools/probes/hold_polymul.py 65536 30 128 3
    echo -n "E one-launch batch 128 level $lv: "; NFLHIP_XCD=1 NFL_POLYMUL_LEVEL=$lv timeout 200 python tools/probes/hold_polymul.py 65536 30 128 3
  done
done
} > gpurun_out/r06_E_incomplete_ab.txt 2>&1
cat gpurun_out/r06_E_incomplete_ab.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06_gputests_c.txt
cat gpurun_out/r06_gputests_c.txt
