#!/bin/bash
# round 4, session c: the header's transform fusion on the real library; whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r4c
mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1 ) 2> $O/pytest_all.time; echo "pytest rc $?" >> $O/pytest_all.log
tail -6 $O/pytest_all.log; tail -3 $O/pytest_all.time
for reps in 2048 16384; do
  NFL_LWE_REPS=$reps NFL_LWE_VERBOSE=1 timeout 600 tests/cpp/resident_test > $O/lwe_poly_p_$reps.json 2> $O/lwe_poly_p_$reps.err
  tail -1 $O/lwe_poly_p_$reps.json; grep "lwe:" $O/lwe_poly_p_$reps.err | head -4
  NFL_HIP_NO_FUSION=1 NFL_LWE_REPS=$reps NFL_LWE_VERBOSE=1 timeout 600 tests/cpp/resident_test > $O/lwe_poly_p_nofusion_$reps.json 2> $O/lwe_poly_p_nofusion_$reps.err
  tail -1 $O/lwe_poly_p_nofusion_$reps.json
done
