#!/bin/bash
# GPU session R of round 3: XCD remap of the n = 65536 pipeline kernel (each twiddle table fetched by ~1.3 of the 8 L2s instead
# of all 8): parity on the n = 65536 tests, then same-box A/B (build/noremap: -DNFLHIP_NO_PIPE_REMAP) on workload E with the
# in-run HBM traffic of both.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "65536 or parity or xcd or golden" > $out/r03r_pytest.txt 2>&1
grep -E "passed|failed|error" $out/r03r_pytest.txt | tail -2
cp nfllib_amd/libnflhip.so /tmp/lib_remap.so
cp build/noremap/nfllib_amd/libnflhip.so /tmp/lib_noremap.so
: > $out/r03r_ab.txt
for rep in 1 2; do
  for v in remap noremap; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    r=$(timeout 600 python bench.py --workload E --steps 60 --warmup 6 --no-cpu-baseline --no-rccl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d.get('extras',{}); r=d['roofline']; print(d['value'], r['frac'], 'traffic x%.3f' % (r['traffic'] / r['algorithmic_bytes_per_launch']) if r.get('traffic') else None, e.get('polymul_b_pretransformed_per_s'), d['config']['self_check'])")
    echo "E $v rep$rep value frac traffic pretransformed ok: $r" >> $out/r03r_ab.txt
  done
done
cp /tmp/lib_remap.so nfllib_amd/libnflhip.so
sort -s -k2,2 $out/r03r_ab.txt
