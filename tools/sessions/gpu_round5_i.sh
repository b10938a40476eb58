#!/bin/bash
# round 5, session i: the poly_p loop after the half-length first runs, 2 048 ... 65 536 iterations
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for reps in 512 1024 1536 2048 3072 4096 16384 65536; do
  echo "== reps $reps (default policy)"
  NFL_LWE_REPS=$reps NFL_LWE_VERBOSE=1 tests/cpp/resident_test 2>&1 | grep "lwe:\|poly_p_enc" | head -3 | grep -v "recorded in [0-9]*\.[0-9]* ms (incl. queue runs), finished after [0-9][0-9][0-9]" | cut -c1-200
  echo "== reps $reps (NFL_HIP_QUEUE_LIMIT=8192: the fixed length of round 4)"
  NFL_LWE_REPS=$reps NFL_HIP_QUEUE_LIMIT=8192 tests/cpp/resident_test 2>&1 | grep "poly_p_enc" | cut -c1-200
done
} > gpurun_out/r05_short_loops_34.txt 2>&1
cat gpurun_out/r05_short_loops_34.txt
python -m pytest tests/test_zz_gpu_deferred_loops.py tests/test_cpp_surface.py -m gpu -q 2>&1 | tail -3
