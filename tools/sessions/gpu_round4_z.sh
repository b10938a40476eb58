#!/bin/bash
# round 4, session z: the recording path's lock as one atomic per acquisition: the loop alone against the real library, then resident_test
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4z
mkdir -p $O
for i in 1 2 3; do build/ab/lwe_record_real 131072 2>&1 | grep "host cost"; done | tee $O/split.txt
for rep in 1 2 3; do for reps in 2048 65536; do
  echo -n "resident_test $reps: " | tee -a $O/ab.txt
  NFL_LWE_REPS=$reps timeout 300 tests/cpp/resident_test 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['lwe_u64_4096_4']; print({k:v for k,v in d.items() if (k.startswith('poly_p_e') or k.startswith('poly_p_d')) and 'eager' not in k or 'launch' in k})" | tee -a $O/ab.txt
done; done
