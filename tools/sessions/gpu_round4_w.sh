#!/bin/bash
# round 4, session w: where the plain-poly_p LWE loop's wall time goes on the real library: the loop alone (tools/hostprof/lwe_record.cpp linked
# against the real library), split into recording loop + final queue run, and its HIP API / kernel trace summary
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4w
mkdir -p $O
for i in 1 2 3; do HOSTPROF_SPLIT=1 build/ab/lwe_record_real 65536 2>&1 | grep -v amdgpu.ids | tail -3; done | tee $O/split.txt
cd /tmp
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/prof_pp -- $GRAFT_REPO_ROOT/build/ab/lwe_record_real 65536 > $O/run.log 2>&1
for f in $(find /tmp/prof_pp -name "*stats*.csv"); do echo "== $f"; head -14 $f | cut -c1-150; done | tee $O/stats.txt
