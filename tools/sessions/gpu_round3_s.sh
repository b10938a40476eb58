#!/bin/bash
# GPU session S of round 3: does the driver's short run (--steps 20 --warmup 5: 15 ms of warm-up, 60 ms timed) sit on the
# power controller's ramp?  The same line with 5 / 50 / 400 warm-up steps, three times each, alternating.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
: > $out/r03s_warmup.txt
for rep in 1 2 3; do
  for w in 5 50 400; do
    r=$(timeout 300 python bench.py --gpus 1 --steps 20 --warmup $w --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])")
    echo "warmup $w rep$rep value ms_per_step kernel_ms: $r" >> $out/r03s_warmup.txt
  done
done
sort -s -k2,2n $out/r03s_warmup.txt
