#!/bin/bash
# GPU session AD of round 3: stand-alone forward transforms of 16384- / 8192-word rows with two rows per workgroup: parity, then
# same-box A/B against the previous commit (bench extras of C and G).
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "16384 or 8192 or parity or golden or fuzz or rows" > $out/r03ad_pytest.txt 2>&1
grep -E "passed|failed|error" $out/r03ad_pytest.txt | tail -2
cp nfllib_amd/libnflhip.so /tmp/lib_new.so
cp build/prev/nfllib_amd/libnflhip.so /tmp/lib_prev.so
: > $out/r03ad_ab.txt
for rep in 1 2; do
  for v in new prev; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    for wl in C G; do
      r=$(timeout 300 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-traffic --no-rccl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d.get('extras',{}); print(d['value'], e.get('ntt_fwd_per_s'), e.get('ntt_fwd_GBs'), e.get('ntt_inv_per_s'), d['config']['self_check'])")
      echo "$wl $v rep$rep value fwd fwd_GBs inv ok: $r" >> $out/r03ad_ab.txt
    done
  done
done
cp /tmp/lib_new.so nfllib_amd/libnflhip.so
sort -s -k1,1 -k2,2 $out/r03ad_ab.txt
