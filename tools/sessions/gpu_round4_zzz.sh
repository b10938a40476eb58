#!/bin/bash
# round 4: the plain-poly_p loop's HIP API summary after the host-side changes (before: gpurun_out/r4w/stats.txt); resident_test figures with the batch's fused methods
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4zzz
mkdir -p $O
out=$GRAFT_REPO_ROOT/gpurun_out
tests/cpp/resident_test | head -1 > $out/r04_final_lwe_poly_p.json
NFL_LWE_REPS=16384 tests/cpp/resident_test | head -1 >> $out/r04_final_lwe_poly_p.json
NFL_LWE_REPS=65536 tests/cpp/resident_test | head -1 >> $out/r04_final_lwe_poly_p.json
cat $out/r04_final_lwe_poly_p.json
for i in 1 2 3; do build/ab/lwe_record_real 65536 2>&1 | grep "host cost"; done
cd /tmp
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/prof_pp -- $GRAFT_REPO_ROOT/build/ab/lwe_record_real 65536 > $O/run.log 2>&1
for f in $(find /tmp/prof_pp -name "*hip_api_stats.csv"); do head -12 $f | cut -c1-150; done | tee $O/hip_api_after.txt
