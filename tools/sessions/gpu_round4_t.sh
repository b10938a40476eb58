#!/bin/bash
# round 4, session t: plain-poly_p LWE loop against the deferred queue's length (host cache locality against launch size)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4t
mkdir -p $O
for rep in 1 2; do
for q in 8192 16384 32768 65536; do
  echo -n "limit $q: " | tee -a $O/queue_limit.txt
  NFL_HIP_QUEUE_LIMIT=$q NFL_LWE_REPS=65536 timeout 300 tests/cpp/resident_test 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['lwe_u64_4096_4']; print({k:v for k,v in d.items() if k.startswith('poly_p_e') or k.startswith('poly_p_d') or 'launch' in k})" | tee -a $O/queue_limit.txt
done; done
