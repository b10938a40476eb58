#!/bin/bash
# round 4, session e: n = 65536 with a pre-transformed operand through the pipeline kernel (b_ntt variant); host-pointer concurrency
export TMPDIR=/tmp
O=gpurun_out/r4e
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_xcd.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python bench.py --workload E --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_E.json 2> $O/bench_E.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4e/bench_E.json'))
e = d['extras']
print('E', d['value'], d['roofline']['frac'], d['roofline']['traffic'] / d['roofline']['algorithmic_bytes_per_launch'], 'pretransformed', e['polymul_b_pretransformed_per_s'],
      'ratio', e['polymul_b_pretransformed_per_s'] / d['value'], 'crt lift', e['crt_lift_per_s'])
PY
