#!/bin/bash
# round 4, session f: n = 65536 pipeline -- chunks per batch x batch size (libraries built with -DNFLHIP_PIPE_CHUNKS=N under build/ab/)
export TMPDIR=/tmp
O=gpurun_out/r4f
mkdir -p $O
cp nfllib_amd/libnflhip.so /tmp/libnflhip_default.so
for lib in default chunks2 chunks6 chunks8; do
  if [ $lib = default ]; then cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so; else cp build/ab/libnflhip_$lib.so nfllib_amd/libnflhip.so; fi
  for batch in 32 64 128; do
    timeout 300 python bench.py --workload E --batch $batch --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-traffic --no-rccl > /tmp/e.json 2>> $O/err.log
    python - $lib $batch <<'PY' | tee -a $O/sweep.txt
import json, sys
d = json.load(open('/tmp/e.json'))
print(sys.argv[1], 'batch', sys.argv[2], 'polymul/s', d['value'], 'frac', d['roofline']['frac'], 'ok', d['config']['self_check'])
PY
  done
done
cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so
