#!/bin/bash
# GPU session D of round 3: tests; same-box A/B of the 128-VGPR row kernels: one butterfly at a time + 9-slot twiddle ring
# ("ring", shipped) against two interleaved butterflies + 5-slot ring ("ringpair", build/alt); the pipelined host-pointer path
set -u
out=gpurun_out
mkdir -p $out
here=$(pwd)
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $out/r03d_pytest.txt
tail -3 $out/r03d_pytest.txt
cp nfllib_amd/libnflhip.so /tmp/lib_ring.so
cp build/alt/nfllib_amd/libnflhip.so /tmp/lib_ringpair.so
: > $out/r03d_ab.txt
for rep in 1 2; do
  for v in ring ringpair; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    for wl in C F G; do
      r=$(timeout 300 python bench.py --workload $wl --steps 60 --warmup 10 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d['extras']; print(d['value'], d['roofline']['kernel_ms'], e['ntt_fwd_per_s'], e['ntt_inv_per_s'], e['polymul_b_pretransformed_per_s'])")
      echo "$v $wl rep$rep value kernel_ms fwd inv pretransformed: $r" >> $out/r03d_ab.txt
    done
  done
done
cp /tmp/lib_ring.so nfllib_amd/libnflhip.so
cat $out/r03d_ab.txt
timeout 600 python bench.py > $out/r03d_bench_B.json 2> $out/r03d_bench_B.err
python -c "
import json; d=json.loads(open('$out/r03d_bench_B.json').readline()); print(d['value'], d['roofline']['frac'], d['extras']['host_pointer_polymul_per_s'])"
