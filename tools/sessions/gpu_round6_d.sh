#!/bin/bash
# round 6, session d: held rates of the long-row / row-resident products at level 0 (complete transforms) and 2 (incomplete),
# alternated on the same box: E (pipeline, one-launch), C, G; parity of the new row-resident kernels first
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_incomplete.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06_incomplete_tests_d.txt
cat gpurun_out/r06_incomplete_tests_d.txt
{
for rep in 1 2; do
  for lv in 0 2; do
    echo -n "E pipeline batch 128 level $lv: "; NFLHIP_XCD=0 NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py 65536 30 128 3
    echo -n "E one-launch batch 128 level $lv: "; NFLHIP_XCD=1 NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py 65536 30 128 3
    echo -n "C batch 2048 level $lv: "; NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py 16384 8 2048 3
    echo -n "G batch 8192 level $lv: "; NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py 8192 2 8192 3
    echo -n "B batch 16384 level $lv: "; NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py 4096 4 16384 3
  done
done
} > gpurun_out/r06_long_rows_incomplete_ab.txt 2>&1
cat gpurun_out/r06_long_rows_incomplete_ab.txt
