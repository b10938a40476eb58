#!/bin/bash
# GPU session P of round 3: "constant x data" operand order in every generated kernel's v_mad_u64_u32 (shipped) against
# "data x constant" (build/noswap: NFL_GEN_SWAP_MAD=0): the whole GPU suite on the shipped build, then same-box A/B per workload.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $out/r03p_pytest.txt 2>&1
grep -E "passed|failed|error" $out/r03p_pytest.txt | tail -2
cp nfllib_amd/libnflhip.so /tmp/lib_swap.so
cp build/noswap/nfllib_amd/libnflhip.so /tmp/lib_noswap.so
: > $out/r03p_ab.txt
for rep in 1 2; do
  for v in swap noswap; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    for wl in B A C F G H T E; do
      r=$(timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-rccl --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'], d['config']['self_check'])")
      echo "$wl $v rep$rep value kernel_ms ok: $r" >> $out/r03p_ab.txt
    done
  done
done
cp /tmp/lib_swap.so nfllib_amd/libnflhip.so
sort -s -k1,1 -k2,2 $out/r03p_ab.txt
