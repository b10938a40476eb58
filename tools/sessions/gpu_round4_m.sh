#!/bin/bash
# round 4, session m: matrix-core CRT lift, 8-wave workgroups (4 waves per SIMD) against the 4-wave version; parity first
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crt" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
cp nfllib_amd/libnflhip.so /tmp/libnflhip_default.so
for v in default crtv2 default crtv2; do
  if [ $v = default ]; then cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so; else cp build/ab/libnflhip_$v.so nfllib_amd/libnflhip.so; fi
  timeout 200 python tools/probes/crt_lift_time.py $v >> $O/ab.txt 2>&1
done
cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so
grep -v amdgpu.ids $O/ab.txt
timeout 300 python tools/probes/crt_lift_sweep.py default 2>&1 | grep -v amdgpu.ids | tee $O/sweep.txt
