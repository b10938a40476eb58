#!/bin/bash
# GPU session T of round 3: exchanges under the arithmetic in the two-operand row products (n = 16384 / 8192: a's butterflies
# first, records held, then b's): parity, then same-box A/B against the serial schedule (build/serial: NFL_GEN_SERIAL_EXCHANGE=1).
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "16384 or 8192 or parity or golden or fuzz or rows" > $out/r03t_pytest.txt 2>&1
grep -E "passed|failed|error" $out/r03t_pytest.txt | tail -2
cp nfllib_amd/libnflhip.so /tmp/lib_split.so
cp build/serial/nfllib_amd/libnflhip.so /tmp/lib_serial.so
: > $out/r03t_ab.txt
for rep in 1 2 3; do
  for v in split serial; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    for wl in C G; do
      r=$(timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-rccl --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'], d['config']['self_check'])")
      echo "$wl $v rep$rep value frac ok: $r" >> $out/r03t_ab.txt
    done
  done
done
cp /tmp/lib_split.so nfllib_amd/libnflhip.so
sort -s -k1,1 -k2,2 $out/r03t_ab.txt
