#!/bin/bash
# GPU session Y of round 3 (closing): the whole GPU suite on the last library, smoke(), the driver's bench line, and the C / G
# evidence files again (their kernels changed after the r03_final session).
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $out/r03y_pytest.txt 2>&1
grep -E "passed|failed|error" $out/r03y_pytest.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r03y_bench_driver_line.json 2> $out/r03y_bench_driver_line.err
python -c "import json; d=json.loads(open('gpurun_out/r03y_bench_driver_line.json').readline()); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'] / d['roofline']['algorithmic_bytes_per_launch'], d['cpu_baseline']['value'], d['roofline']['secondary']['power'])"
for wl in C G; do bash tools/prof_one.sh $wl r03_final > /dev/null 2>&1; python -c "import json; d=json.loads(open('gpurun_out/r03_final_bench_$wl.json').readline()); print('$wl', d['value'], d['roofline']['frac'])"; done
