#!/bin/bash
# round 6, session f: the generated 64-bit wave-per-row kernels (rows of 1024 / 2048 words): parity, then rates against the compiled
# template (NFLHIP_VARIANT=hipcc) and between level 0 / 2 on the same box; LWE demo on those rings; gpu suite
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_incomplete.py tests/test_gpu_rows.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06_rows_u64_tests.txt
cat gpurun_out/r06_rows_u64_tests.txt
{
for rep in 1 2; do
  for cfgs in "1024 2 32768" "2048 2 16384" "1024 8 8192"; do
    set -- $cfgs
    echo -n "compiled k_row n $1 nm $2: "; NFLHIP_VARIANT=hipcc PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py $1 $2 $3 2 2>/dev/null
    for lv in 0 2; do
      echo -n "generated level $lv n $1 nm $2: "; NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py $1 $2 $3 2 2>/dev/null
    done
  done
done
} > gpurun_out/r06_rows_u64_rates.txt 2>&1
cat gpurun_out/r06_rows_u64_rates.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06_gputests_f.txt
cat gpurun_out/r06_gputests_f.txt
