#!/bin/bash
# GPU session AC of round 3 (closing, again): the whole GPU suite on the last library, the soak, and workload F's evidence files
# after the scratch-layout pair of kernels.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $out/r03ac_pytest.txt 2>&1
grep -E "passed|failed|error" $out/r03ac_pytest.txt | tail -2
timeout 600 python tools/soak.py 1500 2>&1 | grep -v amdgpu.ids | grep -c " 0 mismatches"
bash tools/prof_one.sh F r03_closing > /dev/null 2>&1
python -c "import json; d=json.loads(open('gpurun_out/r03_closing_bench_F.json').readline()); r=d['roofline']; print('F', d['value'], r['frac'], r['traffic'] / r['algorithmic_bytes_per_launch'], r['kernel'])"
