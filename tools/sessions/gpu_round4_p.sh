#!/bin/bash
# round 4, session p: CRT projection on the matrix cores -- parity, rate at E, modulus-count sweep (lift + project)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4p
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crt" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 300 python tools/crt_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/crt_bench.txt
timeout 300 python tools/probes/crt_lift_sweep.py default 2>&1 | grep -v amdgpu.ids | tee $O/sweep.txt
