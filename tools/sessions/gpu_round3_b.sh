#!/bin/bash
# GPU session B of round 3: tests; F with the register-resident 32768-word row kernels against round 2's plans; A and B
# after the SGPR-operand fix; GRBM_GUI_ACTIVE on the issue-rate micro-benchmark (what s_memtime counts under load)
set -u
out=gpurun_out
mkdir -p $out
here=$(pwd)
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $out/r03b_pytest.txt
tail -3 $out/r03b_pytest.txt
for wl in F A B; do
  timeout 600 python bench.py --workload $wl > $out/r03b_bench_$wl.json 2> $out/r03b_bench_$wl.err
  cut -c1-260 $out/r03b_bench_$wl.json
done
NFLHIP_ROW32K=0 timeout 600 python bench.py --workload F --no-cpu-baseline > $out/r03b_bench_F_round2_plan.json 2>/dev/null
cut -c1-260 $out/r03b_bench_F_round2_plan.json
rm -rf /tmp/pmc_ub
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_ub -- $here/build/ubench_issue _4 > $here/$out/r03b_ubench_under_pmc.txt 2>&1)
f=$(find /tmp/pmc_ub -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/r03b_pmc_GRBM_GUI_ACTIVE_ubench.csv
