#!/bin/bash
# round 4, session i: ring-mode vs pair-mode map for every fused kernel at 4096 (mode 3 swaps the default), XCD-dealt grid at 16384 x 8
export TMPDIR=/tmp
O=gpurun_out/r4i
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do for grid in 0 3; do
  timeout 300 python tools/lwe_demo.py --plan fused --batch 16384 --reps 20 --fixed-key --grid $grid >> $O/lwe_grid.jsonl 2>> $O/lwe.err
done; done
timeout 600 python tools/lwe_demo.py --degree 16384 --nmoduli 8 --batch 1024 --plan fused --reps 5 --traffic --fixed-key >> $O/lwe_grid.jsonl 2>> $O/lwe.err
python - <<'PY'
import json
for l in open('gpurun_out/r4i/lwe_grid.jsonl'):
    d = json.loads(l); print(d['degree'], 'mode', d['grid'], 'enc', d['encryptions_per_s'], 'dec', d['decryptions_per_s'], d['digest']['dec'], {k: v for k, v in (d.get('traffic') or {}).items() if 'ratio' in k})
PY
