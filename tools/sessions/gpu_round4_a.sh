#!/bin/bash
# round 4, session a: first run of the transform-fused pipelines on the MI355X
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/r4a/pytest_fused.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4a/pytest_fused.log
tail -5 gpurun_out/r4a/pytest_fused.log
for plan in unfused fused; do
  timeout 300 python tools/lwe_demo.py --plan $plan --batch 8192 --reps 10 --fixed-key >> gpurun_out/r4a/lwe.jsonl 2>> gpurun_out/r4a/lwe.err
done
timeout 600 python tools/lwe_demo.py --plan fused --batch 8192 --reps 10 --traffic --fixed-key >> gpurun_out/r4a/lwe_traffic.jsonl 2>> gpurun_out/r4a/lwe.err
cat gpurun_out/r4a/lwe.jsonl gpurun_out/r4a/lwe_traffic.jsonl
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4a/prof_fused -- python $GRAFT_REPO_ROOT/tools/lwe_demo.py --plan fused --batch 8192 --reps 10 --fixed-key > /dev/null 2>&1)
find gpurun_out/r4a/prof_fused -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4a/lwe_fused_kernel_stats.csv
head -8 gpurun_out/r4a/lwe_fused_kernel_stats.csv
rm -rf gpurun_out/r4a/prof_fused
