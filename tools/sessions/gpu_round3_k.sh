#!/bin/bash
# GPU session K of round 3: lane-major twiddle table for the thread-indexed passes (F3 / I1) of every generated 64-bit kernel:
# the whole GPU suite, then every 64-bit workload's bench line (compare with profiles/r03_final_bench_*.json).
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $out/r03k_pytest.txt 2>&1
tail -4 $out/r03k_pytest.txt
: > $out/r03k_bench.txt
for wl in B C F G E; do
  r=$(timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d.get('extras',{}); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], e.get('ntt_fwd_per_s'), e.get('ntt_inv_per_s'), e.get('polymul_b_pretransformed_per_s'))")
  echo "$wl value frac kernel_ms fwd inv pretransformed: $r" >> $out/r03k_bench.txt
done
cat $out/r03k_bench.txt
(smi() { for i in $(seq 1 $2); do rocm-smi -P -c --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done > $out/r03k_smi_$1.jsonl; }; (smi B 36 &); sleep 1; timeout 120 python tools/power_probe.py B 6 2>&1 | grep -v amdgpu.ids > $out/r03k_power_B.txt; sleep 2)
cat $out/r03k_power_B.txt
