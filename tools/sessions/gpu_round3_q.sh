#!/bin/bash
# GPU session Q of round 3: operand order of v_mul_hi_u32 (build/swap_mulhi: NFL_GEN_SWAP_MULHI=1) on the 30-bit and the 62-bit
# product kernels, same box.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cp nfllib_amd/libnflhip.so /tmp/lib_shipped.so
cp build/swap_mulhi/nfllib_amd/libnflhip.so /tmp/lib_mulhi.so
: > $out/r03q_ab.txt
for rep in 1 2 3; do
  for v in shipped mulhi; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    for wl in A B; do
      r=$(timeout 300 python bench.py --workload $wl --steps 150 --warmup 10 --no-cpu-baseline --no-traffic --no-rccl --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'], d['config']['self_check'])")
      echo "$wl $v rep$rep value kernel_ms ok: $r" >> $out/r03q_ab.txt
    done
  done
done
cp /tmp/lib_shipped.so nfllib_amd/libnflhip.so
sort -s -k1,1 -k2,2 $out/r03q_ab.txt
