#!/bin/bash
# GPU session A of round 3: tests, the headline bench, the issue-rate grid with the effective clock, GRBM_GUI_ACTIVE passes
set -u
out=gpurun_out
mkdir -p $out
here=$(pwd)
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $out/r03a_pytest.txt
timeout 600 python bench.py > $out/r03a_bench_B.json 2> $out/r03a_bench_B.err
timeout 300 ./build/ubench_issue > $out/r03_ubench_issue.txt 2>&1
for wl in B A; do
  rm -rf /tmp/pmc_clk_$wl /tmp/kt_$wl
  (cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_clk_$wl -- python $here/bench.py --workload $wl --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --no-rccl > /dev/null 2> $here/$out/r03a_pmc_clk_$wl.err)
  f=$(find /tmp/pmc_clk_$wl -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/r03a_pmc_GRBM_GUI_ACTIVE_$wl.csv
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$wl -- python $here/bench.py --workload $wl --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic --no-rccl > /dev/null 2>> $here/$out/r03a_pmc_clk_$wl.err)
  f=$(find /tmp/kt_$wl -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/r03a_kernel_trace_$wl.csv
  f=$(find /tmp/kt_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/r03a_kernel_stats_$wl.csv
done
tail -3 $out/r03a_pytest.txt
cut -c1-400 $out/r03a_bench_B.json
head -5 $out/r03_ubench_issue.txt
