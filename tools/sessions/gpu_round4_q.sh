#!/bin/bash
# round 4, session q: workload E (u64/65536/30) product against the batch size beyond 128; bench B's side configs (CRT both ways at E)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4q
mkdir -p $O
for batch in 128 192 256 384; do
  timeout 400 python bench.py --workload E --batch $batch --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-traffic --no-rccl > /tmp/e.json 2>> $O/err.log
  python - $batch <<'PY' | tee -a $O/sweep.txt
import json, sys
d = json.load(open('/tmp/e.json'))
print('batch', sys.argv[1], 'polymul/s', d['value'], 'frac', d['roofline']['frac'], 'ok', d['config']['self_check'])
PY
done
timeout 600 python bench.py --workload E --batch 256 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-rccl > $O/bench_E_256.json 2>> $O/err.log
python -c "
import json; d=json.load(open('$O/bench_E_256.json')); print('E 256 with traffic:', d['value'], d['roofline'])"
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_B.json 2>> $O/err.log
python -c "
import json; d=json.load(open('$O/bench_B.json')); print(d['value'], d['roofline']['frac']); print(json.dumps(d['extras']['configs'], indent=1)[:2500])"
tail -5 $O/err.log
