#!/bin/bash
# GPU session M of round 3: split-phase exchanges in the 32768-word row kernels (the LDS batches of one file under the
# arithmetic of the other): parity on every test that touches n = 32768, then same-box A/B against the serial schedule
# (build/serial32k: NFL_GEN_SERIAL_EXCHANGE=1).
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "32768 or xcd or parity or golden or fuzz" > $out/r03m_pytest.txt 2>&1
tail -3 $out/r03m_pytest.txt
cp nfllib_amd/libnflhip.so /tmp/lib_split.so
cp build/serial32k/nfllib_amd/libnflhip.so /tmp/lib_serial.so
: > $out/r03m_ab.txt
for rep in 1 2 3; do
  for v in split serial; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    r=$(timeout 300 python bench.py --workload F --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-rccl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d.get('extras',{}); print(d['value'], d['roofline']['frac'], e.get('ntt_fwd_per_s'), e.get('ntt_inv_per_s'), e.get('polymul_b_pretransformed_per_s'), d['config']['self_check'])")
    echo "$v F rep$rep value frac fwd inv pretransformed ok: $r" >> $out/r03m_ab.txt
  done
done
cp /tmp/lib_split.so nfllib_amd/libnflhip.so
sort -k1,1 -s $out/r03m_ab.txt
