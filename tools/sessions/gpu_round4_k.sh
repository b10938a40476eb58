#!/bin/bash
# round 4, session k: where the time of the matrix-core CRT lift goes -- builds with one phase's arithmetic removed
export TMPDIR=/tmp
O=gpurun_out/r4k
mkdir -p $O
cp nfllib_amd/libnflhip.so /tmp/libnflhip_default.so
for v in default 1 2 4 7; do
  if [ $v = default ]; then cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so; else cp build/ab/libnflhip_crtprobe$v.so nfllib_amd/libnflhip.so; fi
  timeout 200 python tools/probes/crt_lift_time.py $v >> $O/phases.txt 2>&1
done
cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so
cat $O/phases.txt
