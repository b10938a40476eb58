#!/bin/bash
# round 5, session k: the metric kernel's rocprofv3 summary without the side configurations (config D's 2^17-polynomial launches
# carry the same kernel name and would pollute the average), and the big-delta family's rates
cd "$(dirname "$0")/../.."
here=$(pwd); export TMPDIR=/tmp PYTHONPATH=$here
mkdir -p gpurun_out
rm -rf /tmp/prof_B
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_B -- python $here/bench.py --workload B --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-traffic --no-side-configs > $here/gpurun_out/r05_final_bench_B_under_rocprof.json 2>/dev/null)
cp $(find /tmp/prof_B -name "*kernel_stats.csv" | head -1) gpurun_out/r05_final_kernel_stats_B.csv
head -3 gpurun_out/r05_final_kernel_stats_B.csv | cut -c1-160
cut -c1-200 gpurun_out/r05_final_bench_B_under_rocprof.json
{
echo "contexts past the 92nd modulus (general-modulus kernels) against the delta-form kernels: products per second PER ROW (polymul/s x moduli)"
for cfg in "4096 92 256" "4096 95 256" "1024 92 1024" "1024 94 1024" "16384 92 32" "16384 96 32" "65536 92 4" "65536 94 4"; do
  set -- $cfg
  python tools/probes/hold_polymul.py $1 $2 $3 1.0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n', d['n'], 'moduli', d['nm'], 'batch', d['batch'], 'polymul/s', d['polymul_per_s'], 'rows/s', round(d['polymul_per_s']*d['nm']), 'frac', d['frac'])"
done
} > gpurun_out/r05_big_delta_rates.txt 2>&1
cat gpurun_out/r05_big_delta_rates.txt
