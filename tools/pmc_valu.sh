#!/bin/bash
# VALU utilisation of one workload's product kernel: separate rocprofv3 --pmc passes (no trace domains), mean per launch.
#   tools/pmc_valu.sh A "k_row<"        (workload, kernel-name substring)
cd /tmp && export TMPDIR=/tmp
W=$1; K=$2; R=${GRAFT_REPO_ROOT:-/root/repo}
dirs=""
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS"; do
  d=/tmp/pmcv_$(echo $c | tr ' ' '_'); rm -rf $d
  rocprofv3 --pmc $c --output-format csv -d $d -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-traffic > /dev/null 2>&1 || echo "pass $c failed"
  dirs="$dirs $d"
done
python $R/tools/pmc_sq.py "$K" $dirs
