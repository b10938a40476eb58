#!/bin/bash
# GPU session X of round 3 (the same after the forward reads of the two-operand products became progressive too): progressive exchanges in the 16384- / 8192-word kernels (words written to the LDS out of a pass's
# last stage, reads issued in the consumer's order with per-butterfly waits): parity, then same-box A/B against the previous commit.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "16384 or 8192 or parity or golden or fuzz or rows" > $out/r03x_pytest.txt 2>&1
grep -E "passed|failed|error" $out/r03x_pytest.txt | tail -2
cp nfllib_amd/libnflhip.so /tmp/lib_new.so
cp build/prev/nfllib_amd/libnflhip.so /tmp/lib_prev.so
: > $out/r03x_ab.txt
for rep in 1 2; do
  for v in new prev; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    for wl in C G; do
      r=$(timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-rccl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d.get('extras',{}); print(d['value'], d['roofline']['frac'], e.get('ntt_fwd_per_s'), e.get('ntt_inv_per_s'), e.get('polymul_b_pretransformed_per_s'), d['config']['self_check'])")
      echo "$wl $v rep$rep value frac fwd inv pretransformed ok: $r" >> $out/r03x_ab.txt
    done
  done
done
cp /tmp/lib_new.so nfllib_amd/libnflhip.so
sort -s -k1,1 -k2,2 $out/r03x_ab.txt
