#!/bin/bash
# round 4, session x: the sampler-argument validation remembered per distinct tuple (was: only the last one -> two validating C-ABI calls per
# encryption of the LWE loop): the loop alone against the real library, then tests/cpp/resident_test against the binary built before the change
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4x
mkdir -p $O
for i in 1 2 3; do HOSTPROF_SPLIT=1 build/ab/lwe_record_real 65536 2>&1 | grep -v amdgpu.ids | tail -2; done | tee $O/split.txt
for rep in 1 2 3; do for v in tests/cpp/resident_test build/ab/resident_test_prejoin; do
  for reps in 2048 65536; do
  echo -n "$v $reps: " | tee -a $O/ab.txt
  NFL_LWE_REPS=$reps timeout 300 $v 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['lwe_u64_4096_4']; print({k:v for k,v in d.items() if (k.startswith('poly_p_e') or k.startswith('poly_p_d') or k.startswith('device')) and 'eager' not in k or 'launch' in k})" | tee -a $O/ab.txt
done; done; done
