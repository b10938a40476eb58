#!/bin/bash
# round 6, session aa: soak of the third session's kernels -- tools/soak.py (determinism over repeated launches, odd batches), the
# randomised GPU tests under four more seeds, then the whole -m gpu suite on the tree
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 1500 python tools/soak.py 150 2>&1 | grep -v amdgpu.ids
for seed in 101 20261001 77777 424242; do
  echo "== NFL_FUZZ_SEED=$seed"
  NFL_FUZZ_SEED=$seed timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_fused.py tests/test_gpu_incomplete.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
done
} > gpurun_out/r06_soak3.txt 2>&1
cat gpurun_out/r06_soak3.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06_gputests_aa.txt
cat gpurun_out/r06_gputests_aa.txt
