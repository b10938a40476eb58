#!/bin/bash
# One workload's round-end evidence (GPU box): its bench line (roofline with in-run traffic, cpu_baseline) and the rocprofv3
# kernel summary of the same command.   bash tools/prof_one.sh A [tag]   -> gpurun_out/<tag>_bench_A.json, ..._kernel_stats_A.csv
set -u
wl=${1:-A}
tag=${2:-r02_final}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
here=$(pwd)
python bench.py --workload $wl 2>/dev/null | tail -1 > $out/${tag}_bench_${wl}.json
rm -rf /tmp/prof_$wl
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python $here/bench.py --workload $wl --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-traffic --no-side-configs > $here/$out/${tag}_bench_${wl}_under_rocprof.json 2>/dev/null)
f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats_${wl}.csv
rm -rf /tmp/prof_$wl
cut -c1-300 $out/${tag}_bench_${wl}.json
