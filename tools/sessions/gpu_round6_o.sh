#!/bin/bash
# round 6, session o: the queue's own thread after the pin-indexed rewrite (a run touches no payload)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
make -s -C tests/cpp resident_test deferred_loops deferred_fuzz deferred_threads deferred_edges 2>&1 | tail -3
{
echo "# encryptions/s, decryptions/s of the poly_p loop (tests/cpp/resident_test): queue thread on / off, minimum run length (records)"
for reps in 16384 2048 4096 1024; do
  for cfg in "0 1024 8192" "1 1024 8192"; do
    set -- $cfg
    for rep in 1 2 3; do
      NFL_HIP_QUEUE_THREAD=$1 NFL_HIP_QUEUE_MIN=$2 NFL_HIP_QUEUE_LIMIT=$3 NFL_LWE_REPS=$reps timeout 300 tests/cpp/resident_test 2>/dev/null | head -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['lwe_u64_4096_4']
print('reps $reps thread $1 min_run $2 limit $3: poly_p %.3f M enc/s %.3f M dec/s; batch fused %.3f M enc/s; launches %s for %s operations' % (d['poly_p_encryptions_per_s'] / 1e6, d['poly_p_decryptions_per_s'] / 1e6, d['device_batch_fused_encryptions_per_s'] / 1e6, d.get('launches_they_became'), d.get('deferred_operations')))"
    done
  done
done
for t in 0 1; do
echo "## per-run times, thread $t, min_run 100000"
NFL_HIP_QUEUE_STATS=1 NFL_HIP_QUEUE_THREAD=$t NFL_HIP_QUEUE_MIN=100000 NFL_LWE_REPS=16384 tests/cpp/resident_test 2>&1 >/dev/null | grep "queue run" | awk '$4 > 1000' | sed -n 12,17p
done
echo "# correctness on the real library, queue thread on (default)"
tests/cpp/deferred_loops 300 | tail -1
NFL_HIP_QUEUE_MIN=64 tests/cpp/deferred_loops 300 | tail -1
tests/cpp/deferred_fuzz 40 2024 | tail -1
NFL_HIP_QUEUE_MIN=32 tests/cpp/deferred_fuzz 40 7 | tail -1
NFL_HIP_QUEUE_LIMIT=37 tests/cpp/deferred_threads 6 400 | tail -1
tests/cpp/deferred_edges | tail -1
} > gpurun_out/r06_queue_thread2.txt 2>&1
cat gpurun_out/r06_queue_thread2.txt
