#!/bin/bash
# round 4, session h: A/B of the ring-mode (128 VGPR, four workgroups per CU) inverse pipeline at 4096 against the pair-mode one
export TMPDIR=/tmp
O=gpurun_out/r4h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "both_grids or multiply_subtract" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do for grid in 0 3; do
  timeout 300 python tools/lwe_demo.py --plan fused --batch 16384 --reps 20 --fixed-key --grid $grid >> $O/lwe_grid.jsonl 2>> $O/lwe.err
done; done
python - <<'PY'
import json
for l in open('gpurun_out/r4h/lwe_grid.jsonl'):
    d = json.loads(l); print('grid', d['grid'], 'enc', d['encryptions_per_s'], 'dec', d['decryptions_per_s'], d['digest']['dec'])
PY
