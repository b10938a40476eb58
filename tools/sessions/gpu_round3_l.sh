#!/bin/bash
# GPU session L of round 3: same-box A/B of the lane-major twiddle table (shipped) against the natural-order table
# (build/natural_tw: NFL_GEN_NATURAL_TWIDDLES=1 + -DNFLHIP_NATURAL_TWIDDLES), products and stand-alone transforms, two rounds.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cp nfllib_amd/libnflhip.so /tmp/lib_lm.so
cp build/natural_tw/nfllib_amd/libnflhip.so /tmp/lib_nat.so
: > $out/r03l_ab.txt
for rep in 1 2; do
  for v in lm nat; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    for wl in B C F G E; do
      r=$(timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-traffic --no-rccl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d.get('extras',{}); print(d['value'], d['roofline']['kernel_ms'], e.get('ntt_fwd_per_s'), e.get('ntt_inv_per_s'), e.get('polymul_b_pretransformed_per_s'), d['config']['self_check'])")
      echo "$v $wl rep$rep value kernel_ms fwd inv pretransformed ok: $r" >> $out/r03l_ab.txt
    done
  done
done
cp /tmp/lib_lm.so nfllib_amd/libnflhip.so
sort -k2,2 -k1,1 -s $out/r03l_ab.txt
