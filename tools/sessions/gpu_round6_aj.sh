#!/bin/bash
# round 6, session aj: soak of the FINAL library (warm-up, pinned staging, token flags on top of the session's kernels): tools/soak.py, the C++
# surface / reference programs / deferred loops three times over, the randomised tests under two more seeds
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 1500 python tools/soak.py 100 2>&1 | grep -v amdgpu.ids | tail -4
for rep in 1 2 3; do
  timeout 900 python -m pytest tests/test_cpp_surface.py tests/test_reference_programs.py tests/test_zz_gpu_deferred_loops.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -2
done
for seed in 31337 987654321; do
  echo "== NFL_FUZZ_SEED=$seed"
  NFL_FUZZ_SEED=$seed timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_fused.py tests/test_cpp_surface.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -2
done
} > gpurun_out/r06_soak4.txt 2>&1
cat gpurun_out/r06_soak4.txt
