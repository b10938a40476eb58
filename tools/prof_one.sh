set -u
tag=r02_final
out=gpurun_out
export TMPDIR=/tmp
here=$(pwd)
python bench.py --workload A 2>/dev/null | tail -1 > $out/${tag}_bench_A.json
rm -rf /tmp/prof_A
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_A -- python $here/bench.py --workload A --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-traffic > $here/$out/${tag}_bench_A_under_rocprof.json 2>/dev/null)
f=$(find /tmp/prof_A -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats_A.csv
cat $out/${tag}_bench_A.json | cut -c1-400
