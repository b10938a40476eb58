#!/bin/bash
# round 4, last session: the whole GPU suite on the final header + library; the plain-poly_p LWE figures of the round-end set again
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4zz
mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1 ) 2> $O/pytest_all.time; echo "pytest rc $?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log; tail -3 $O/pytest_all.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
out=$GRAFT_REPO_ROOT/gpurun_out
tests/cpp/resident_test | head -1 > $out/r04_final_lwe_poly_p.json
NFL_LWE_REPS=16384 tests/cpp/resident_test | head -1 >> $out/r04_final_lwe_poly_p.json
NFL_LWE_REPS=65536 tests/cpp/resident_test | head -1 >> $out/r04_final_lwe_poly_p.json
NFL_HIP_NO_FUSION=1 NFL_LWE_REPS=16384 tests/cpp/resident_test | head -1 > $out/r04_final_lwe_poly_p_nofusion.json
cat $out/r04_final_lwe_poly_p.json $out/r04_final_lwe_poly_p_nofusion.json
