#!/bin/bash
# round 6, session ab: the deferred queue's maximal run length (NFL_HIP_QUEUE_LIMIT, records per run; the queue's own thread on) on the LWE
# loop written with plain poly_p operators (tests/cpp/resident_test): longer runs = larger launches (a run of 8 192 records is 1 638
# encryptions = 2.1 rounds of the chip's wave slots), later start.  Three repetitions each.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for reps in 2048 16384 65536; do
  for lim in 8192 16384 32768 65536; do
    for r in 1 2 3; do
      NFL_HIP_QUEUE_LIMIT=$lim NFL_LWE_REPS=$reps timeout 300 tests/cpp/resident_test 2>/dev/null | head -1 | python -c "
import sys, json
d = list(json.loads(sys.stdin.read()).values())[0]
print('reps $reps limit $lim: poly_p %.3f M enc/s %.3f M dec/s; batch fused %.3f M enc/s; launches %s for %s operations' % (d['poly_p_encryptions_per_s'] / 1e6, d['poly_p_decryptions_per_s'] / 1e6, d['device_batch_fused_encryptions_per_s'] / 1e6, d.get('launches_they_became'), d.get('deferred_operations')))"
    done
  done
done
} > gpurun_out/r06_queue_limit.txt 2>&1
cat gpurun_out/r06_queue_limit.txt
