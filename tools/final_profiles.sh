#!/bin/bash
# Round-end evidence (GPU box): the bench line of every workload and the rocprofv3 kernel summary of the metric run.
# Usage: bash tools/final_profiles.sh <tag>   -> gpurun_out/<tag>_bench_{A,B,C,E}.json, gpurun_out/<tag>_kernel_stats_B.csv
set -u
tag=${1:-r01_final}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for wl in B A C E; do
  python bench.py --workload $wl 2>/dev/null | grep '^{' | tail -1 > $out/${tag}_bench_${wl}.json
done
rm -rf $out/prof_B
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_B -- python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $out/${tag}_bench_B_under_rocprof.json 2>/dev/null
f=$(find $out/prof_B -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats_B.csv
rm -rf $out/prof_B
python tools/lwe_demo.py 2>/dev/null > $out/${tag}_lwe.jsonl
python tools/lwe_demo.py --degree 16384 --nmoduli 8 --batch 512 2>/dev/null >> $out/${tag}_lwe.jsonl
python tools/lwe_demo.py --degree 1024 --nmoduli 2 --batch 65536 2>/dev/null >> $out/${tag}_lwe.jsonl
ls -la $out | tail -12
