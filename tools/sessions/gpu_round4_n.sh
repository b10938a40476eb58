#!/bin/bash
# round 4, session n: s_memtime stamps per phase of the matrix-core CRT lift (probe build, printf from workgroup 3)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4n
mkdir -p $O
cp nfllib_amd/libnflhip.so /tmp/libnflhip_default.so
cp build/ab/libnflhip_crtstamp.so nfllib_amd/libnflhip.so
timeout 200 python tools/probes/crt_lift_once.py 64 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so
