#!/bin/bash
# round 4, session y: queue length again, now that the validating calls are gone (lwe_record against the real library; resident_test)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4y
mkdir -p $O
for q in 4096 8192 16384 32768 65536; do
  echo -n "limit $q: " | tee -a $O/limit.txt
  NFL_HIP_QUEUE_LIMIT=$q build/ab/lwe_record_real 131072 2>&1 | grep "host cost" | tee -a $O/limit.txt
done
for q in 8192 16384 32768; do
  echo -n "resident_test limit $q: " | tee -a $O/limit.txt
  NFL_HIP_QUEUE_LIMIT=$q NFL_LWE_REPS=65536 timeout 300 tests/cpp/resident_test 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['lwe_u64_4096_4']; print({k:v for k,v in d.items() if (k.startswith('poly_p_e') or k.startswith('poly_p_d')) and 'eager' not in k or 'launch' in k})" | tee -a $O/limit.txt
done
