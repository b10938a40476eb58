#!/bin/bash
# round 4, session d: CHECK_STRICTMOD / archive tests, reftests with the range assertion live, grid A/B of the fused kernels, poly_p LWE
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
timeout 1500 python -m pytest tests/test_cpp_surface.py tests/test_reference_programs.py tests/test_gpu_fused.py tests/test_zz_gpu_deferred_loops.py tests/test_abi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for grid in 1 2; do for rep in 1 2; do
  timeout 300 python tools/lwe_demo.py --plan fused --batch 16384 --reps 10 --fixed-key --grid $grid >> $O/lwe_grid.jsonl 2>> $O/lwe.err
done; done
python - <<'PY'
import json
for l in open('gpurun_out/r4d/lwe_grid.jsonl'):
    d = json.loads(l); print('grid', d['grid'], 'enc', d['encryptions_per_s'], 'dec', d['decryptions_per_s'])
PY
for reps in 2048 16384; do
  NFL_LWE_REPS=$reps NFL_LWE_VERBOSE=1 timeout 600 tests/cpp/resident_test > $O/lwe_poly_p_$reps.json 2> $O/lwe_poly_p_$reps.err
  grep "^{" $O/lwe_poly_p_$reps.json; grep "lwe:" $O/lwe_poly_p_$reps.err | head -2
done
