#!/bin/bash
# round 4, session s: where the matrix-core CRT kernels overtake the VALU kernels, both directions, same box (two builds: GEMM never / GEMM from 10-12 moduli)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4s
mkdir -p $O
cp nfllib_amd/libnflhip.so /tmp/libnflhip_default.so
for v in nomfma allmfma; do
  cp build/ab/libnflhip_$v.so nfllib_amd/libnflhip.so
  timeout 300 python tools/probes/crt_lift_sweep.py $v 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.txt
done
cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so
