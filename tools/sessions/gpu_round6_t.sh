#!/bin/bash
# round 6, session t: soak of the deferred queue with its own thread on the real library: random programs, loop shapes and six recording
# threads under many (seed, first hand-over, limit) combinations; every program compares deferred with immediate execution word for word
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
make -s -C tests/cpp deferred_loops deferred_fuzz deferred_threads deferred_edges resident_test 2>&1 | tail -3
{
fail=0; runs=0
for seed in $(seq 101 130); do
  for cfg in "1 1024 8192" "1 16 97" "1 5 40" "1 64 512" "0 1024 8192" "1 1 3"; do
    set -- $cfg
    out=$(NFL_HIP_QUEUE_THREAD=$1 NFL_HIP_QUEUE_MIN=$2 NFL_HIP_QUEUE_LIMIT=$3 timeout 300 tests/cpp/deferred_fuzz 12 $seed 2>&1 | tail -1)
    runs=$((runs+1))
    if [ "$out" != "all checks passed" ]; then fail=$((fail+1)); echo "FUZZ seed $seed cfg $cfg: $out"; fi
  done
done
echo "deferred_fuzz: $runs runs, $fail failures"
fail=0; runs=0
for cfg in "1 1024 8192" "1 16 97" "1 5 40" "1 64 512" "1 7 50" "1 1 3" "1 300 1000"; do
  set -- $cfg
  for n in 300 777; do
    out=$(NFL_HIP_QUEUE_THREAD=$1 NFL_HIP_QUEUE_MIN=$2 NFL_HIP_QUEUE_LIMIT=$3 timeout 300 tests/cpp/deferred_loops $n 2>&1 | tail -1)
    runs=$((runs+1))
    if [ "$out" != "all checks passed" ]; then fail=$((fail+1)); echo "LOOPS $n cfg $cfg: $out"; fi
  done
done
echo "deferred_loops: $runs runs, $fail failures"
fail=0; runs=0
for cfg in "1 16 37" "1 5 40" "1 1024 8192" "1 64 300"; do
  set -- $cfg
  for rep in 1 2 3 4 5; do
    out=$(NFL_HIP_QUEUE_THREAD=$1 NFL_HIP_QUEUE_MIN=$2 NFL_HIP_QUEUE_LIMIT=$3 timeout 300 tests/cpp/deferred_threads 6 400 2>&1 | tail -1)
    runs=$((runs+1))
    if [ "$out" != "all checks passed" ]; then fail=$((fail+1)); echo "THREADS cfg $cfg: $out"; fi
  done
done
echo "deferred_threads (six recording threads + the queue's): $runs runs, $fail failures"
for rep in 1 2 3 4 5 6 7 8; do NFL_LWE_REPS=$((1000 + 777 * rep)) tests/cpp/resident_test 2>&1 | grep -c "all checks passed"; done | sort | uniq -c | sed 's/^/resident_test (LWE demo, digests vs the batch API) passes: /'
} > gpurun_out/r06_queue_soak.txt 2>&1
cat gpurun_out/r06_queue_soak.txt
