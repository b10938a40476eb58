#!/bin/bash
# GPU session AB of round 3: the composed 32768-word product with b in a scratch layout of its own ([block][pair][thread]: coalesced on both sides, no transposes in the forward kernel); (old text:
# first stage consumes it (butterflies start when two loads have landed): parity, then same-box A/B against the previous commit.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "32768 or xcd or parity or golden or fuzz" > $out/r03ab_pytest.txt 2>&1
grep -E "passed|failed|error" $out/r03ab_pytest.txt | tail -2
cp nfllib_amd/libnflhip.so /tmp/lib_new.so
cp build/prev/nfllib_amd/libnflhip.so /tmp/lib_prev.so
: > $out/r03ab_ab.txt
for rep in 1 2 3; do
  for v in new prev; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    r=$(timeout 300 python bench.py --workload F --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-rccl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d.get('extras',{}); print(d['value'], d['roofline']['frac'], e.get('ntt_fwd_per_s'), e.get('ntt_inv_per_s'), e.get('polymul_b_pretransformed_per_s'), d['config']['self_check'])")
    echo "F $v rep$rep value frac fwd inv pretransformed ok: $r" >> $out/r03ab_ab.txt
  done
done
cp /tmp/lib_new.so nfllib_amd/libnflhip.so
sort -s -k2,2 $out/r03ab_ab.txt
