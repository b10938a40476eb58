#!/bin/bash
# GPU session V of round 3: the high-dword chain of a butterfly starts with a product of which only the low dword is ever used:
# v_mul_lo_u32 instead of v_mad_u64_u32 (constant x data / data x constant), metric kernel, same box.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cp nfllib_amd/libnflhip.so /tmp/lib_shipped.so
: > $out/r03v_ab.txt
for rep in 1 2 3; do
  for v in shipped mullo_cd mullo_dc; do
    lib=/tmp/lib_shipped.so; [ $v != shipped ] && lib=build/$v/nfllib_amd/libnflhip.so
    timeout 120 python tools/ab_probe.py $lib 3 2>&1 | grep -v amdgpu.ids >> $out/r03v_ab.txt
    sleep 1
  done
done
sort -s -k1,1 $out/r03v_ab.txt
