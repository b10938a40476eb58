#!/bin/bash
# round 4, session b: sampler with the LDS table + packed stores, XCD-dealt forward kernels, decrypt with early twiddles; bench line
export TMPDIR=/tmp
O=gpurun_out/r4b
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_samplers.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
for plan in unfused fused; do
  timeout 300 python tools/lwe_demo.py --plan $plan --batch 8192 --reps 10 --fixed-key >> $O/lwe.jsonl 2>> $O/lwe.err
done
timeout 300 python tools/lwe_demo.py --plan fused --batch 32768 --reps 10 --fixed-key >> $O/lwe.jsonl 2>> $O/lwe.err
timeout 600 python tools/lwe_demo.py --plan fused --batch 8192 --reps 10 --traffic --fixed-key >> $O/lwe_traffic.jsonl 2>> $O/lwe.err
cat $O/lwe.jsonl $O/lwe_traffic.jsonl
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_fused -- python $GRAFT_REPO_ROOT/tools/lwe_demo.py --plan fused --batch 8192 --reps 10 --fixed-key > /dev/null 2>&1)
find $O/prof_fused -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/lwe_fused_kernel_stats.csv
head -5 $O/lwe_fused_kernel_stats.csv
rm -rf $O/prof_fused
timeout 1500 python -m pytest tests/test_bench_contract.py -x -q -m gpu > $O/pytest_bench.log 2>&1; echo "pytest rc $?" >> $O/pytest_bench.log
tail -15 $O/pytest_bench.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_B.json 2> $O/bench_B.err ) 2> $O/bench_B.time
cat $O/bench_B.json; tail -3 $O/bench_B.time
