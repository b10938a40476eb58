#!/bin/bash
# round 4, session g: the fused pipelines on the row-resident kernels (rows of 8192 / 16384 words)
export TMPDIR=/tmp
O=gpurun_out/r4g
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_abi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
for shape in "16384 8 1024" "8192 2 8192" "4096 4 16384"; do
  set -- $shape
  for plan in unfused fused; do
    timeout 300 python tools/lwe_demo.py --degree $1 --nmoduli $2 --batch $3 --plan $plan --reps 10 --fixed-key >> $O/lwe.jsonl 2>> $O/lwe.err
  done
done
timeout 600 python tools/lwe_demo.py --degree 16384 --nmoduli 8 --batch 1024 --plan fused --reps 5 --traffic --fixed-key >> $O/lwe_traffic_16384.jsonl 2>> $O/lwe.err
python - <<'PY'
import json
for f in ('gpurun_out/r4g/lwe.jsonl', 'gpurun_out/r4g/lwe_traffic_16384.jsonl'):
    for l in open(f):
        d = json.loads(l)
        print(d['degree'], d['nmoduli'], d['plan'], 'enc', d['encryptions_per_s'], 'dec', d['decryptions_per_s'], d['decrypts_to_zero'], d['digest'],
              {k: v for k, v in (d.get('traffic') or {}).items() if 'ratio' in k})
PY
