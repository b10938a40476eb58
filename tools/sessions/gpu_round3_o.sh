#!/bin/bash
# GPU session O of round 3: which multiply-adds care about operand position?  shipped / all swapped / only those with an SGPR
# second factor (delta, the 2^62 term) / only those with a VGPR second factor (twiddle dwords).  Metric kernel, same box.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cp nfllib_amd/libnflhip.so /tmp/lib_shipped.so
: > $out/r03o_swap_ab.txt
for rep in 1 2 3; do
  for v in shipped swap_mad swap_sgpr swap_vgpr; do
    lib=/tmp/lib_shipped.so; [ $v != shipped ] && lib=build/$v/nfllib_amd/libnflhip.so
    timeout 120 python tools/ab_probe.py $lib 3 2>&1 | grep -v amdgpu.ids >> $out/r03o_swap_ab.txt
    sleep 1
  done
done
sort -s -k1,1 $out/r03o_swap_ab.txt
