#!/bin/bash
# round 4, session v: transforms joined into the producing record (detail::lazy::join_transform): the plain-poly_p LWE loop with and without, same box;
# the deferred-queue programs on the real library
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4v
mkdir -p $O
for t in deferred_loops deferred_fuzz deferred_edges deferred_product; do timeout 300 tests/cpp/$t > $O/$t.log 2>&1; echo "$t rc $?"; done
for rep in 1 2 3; do for v in tests/cpp/resident_test build/ab/resident_test_prejoin; do
  echo -n "$v: " | tee -a $O/join.txt
  NFL_LWE_REPS=65536 timeout 300 $v 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['lwe_u64_4096_4']; print({k:v for k,v in d.items() if k.startswith('poly_p_e') and 'eager' not in k or k.startswith('poly_p_d') and 'eager' not in k or 'launch' in k or 'operations' in k})" | tee -a $O/join.txt
done; done
timeout 900 python -m pytest tests/test_zz_gpu_deferred_loops.py tests/test_cpp_surface.py tests/test_reference_programs.py -x -q -m gpu 2>&1 | tail -3
