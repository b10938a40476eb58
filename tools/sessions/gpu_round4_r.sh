#!/bin/bash
# round 4, session r: matrix-core CRT lift at three workgroups per CU (2 N-tiles x 2 M-tiles per wave, 163 VGPRs) against the two-workgroup kernel; parity, stamps, sweep
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4r
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crt" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
cp nfllib_amd/libnflhip.so /tmp/libnflhip_default.so
for v in default crtv5 default crtv5; do
  if [ $v = default ]; then cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so; else cp build/ab/libnflhip_$v.so nfllib_amd/libnflhip.so; fi
  timeout 200 python tools/probes/crt_lift_time.py $v >> $O/ab.txt 2>&1
done
grep -v amdgpu.ids $O/ab.txt
cp build/ab/libnflhip_crtstamp.so nfllib_amd/libnflhip.so
timeout 200 python tools/probes/crt_lift_once.py 64 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/stamps.txt
cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so
timeout 300 python tools/probes/crt_lift_sweep.py default 2>&1 | grep -v amdgpu.ids | tee $O/sweep.txt
