#!/bin/bash
# round 6, session s: nflhip_sample_gauss_small_multi_dev (x, e0, e1 of an LWE encryption in one launch): parity, then the LWE rates
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_samplers.py tests/test_gpu_fused.py tests/test_cpp_surface.py tests/test_zz_gpu_deferred_loops.py -m gpu -x -q 2>&1 | tail -6
make -s -C tests/cpp resident_test 2>&1 | tail -3
for reps in 16384 2048; do
  for t in 0 1; do
    for rep in 1 2 3; do
      NFL_HIP_QUEUE_THREAD=$t NFL_LWE_REPS=$reps timeout 300 tests/cpp/resident_test 2>/dev/null | head -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['lwe_u64_4096_4']
print('reps $reps thread $t: poly_p %.3f M enc/s %.3f M dec/s; batch fused %.3f M enc/s %.3f M dec/s; launches %s for %s operations' % (d['poly_p_encryptions_per_s'] / 1e6, d['poly_p_decryptions_per_s'] / 1e6, d['device_batch_fused_encryptions_per_s'] / 1e6, d['device_batch_fused_decryptions_per_s'] / 1e6, d.get('launches_they_became'), d.get('deferred_operations')))"
    done
  done
done
PYTHONPATH=$(pwd) timeout 600 python tools/lwe_demo.py 2>&1 | tail -12
} > gpurun_out/r06_multi_draw.txt 2>&1
cat gpurun_out/r06_multi_draw.txt
