#!/bin/bash
# Round-end evidence (GPU box): the bench line of every workload (each carries roofline incl. the HBM traffic measured
# in the same run, and cpu_baseline) and the rocprofv3 kernel summaries of the same commands.
# Usage: bash tools/final_profiles.sh <tag>   -> gpurun_out/<tag>_bench_<W>.json, gpurun_out/<tag>_kernel_stats_<W>.csv, ...
set -u
tag=${1:-r04_final}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for wl in B A C E F G D H T; do
  python bench.py --workload $wl 2>/dev/null | tail -1 > $out/${tag}_bench_${wl}.json
done
here=$(pwd)
for wl in B A C E F; do
  rm -rf /tmp/prof_$wl
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python $here/bench.py --workload $wl --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-traffic --no-side-configs > $here/$out/${tag}_bench_${wl}_under_rocprof.json 2>/dev/null)
  f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats_${wl}.csv
  rm -rf /tmp/prof_$wl
done
# stand-alone transforms of the metric shape
for mode in fwd inv; do
  rm -rf /tmp/prof_t
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -- python $here/tools/ntt_only.py $mode > /dev/null 2>&1)
  f=$(find /tmp/prof_t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats_transform_${mode}.csv
done
rm -rf /tmp/prof_t
tests/cpp/resident_test | head -1 > $out/${tag}_lwe_poly_p.json
NFL_LWE_REPS=16384 tests/cpp/resident_test | head -1 >> $out/${tag}_lwe_poly_p.json   # long loop: queue runs overlap recording
NFL_HIP_NO_FUSION=1 NFL_LWE_REPS=16384 tests/cpp/resident_test | head -1 > $out/${tag}_lwe_poly_p_nofusion.json
# the LWE demo on resident batches: operator by operator and through the transform-fused pipelines (same keystreams)
rm -f $out/${tag}_lwe.jsonl
for shape in "4096 4 16384" "8192 2 8192" "16384 8 1024" "32768 2 1024" "1024 2 65536"; do
  set -- $shape
  for plan in unfused fused; do
    python tools/lwe_demo.py --degree $1 --nmoduli $2 --batch $3 --plan $plan --reps 10 --fixed-key 2>/dev/null >> $out/${tag}_lwe.jsonl
  done
done
python tools/lwe_demo.py --batch 8192 --reps 10 --traffic --fixed-key 2>/dev/null > $out/${tag}_lwe_traffic.json
rm -rf /tmp/prof_lwe
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lwe -- python $here/tools/lwe_demo.py --batch 8192 --reps 10 --fixed-key > /dev/null 2>&1)
f=$(find /tmp/prof_lwe -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats_lwe_fused.csv
rm -rf /tmp/prof_lwe
# CRT both ways (tools/crt_bench.py: B, C on the VALU kernels, E on the matrix cores): rates, kernel summary, and the HBM
# traffic of the two GEMM kernels per launch of 16 polynomials (503 316 480 algorithmic bytes) from separate counter passes
python tools/crt_bench.py 2>/dev/null > $out/${tag}_crt_bench.txt
rm -rf /tmp/prof_crt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_crt -- python $here/tools/crt_bench.py > /dev/null 2>&1)
f=$(find /tmp/prof_crt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_kernel_stats_crt.csv
rm -rf /tmp/prof_crt
dirs=""
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_crt_$c
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_crt_$c -- python $here/tools/crt_bench.py > /dev/null 2>&1)
  dirs="$dirs /tmp/pmc_crt_$c"
done
{ echo "# mean per launch (u64/65536/30 x 16): FETCH_SIZE in KiB, to be doubled on gfx950; WRITE_SIZE in KiB"; python tools/pmc_sq.py k_crt_lift_mfma $dirs | sed 's/^/lift    /'; python tools/pmc_sq.py k_crt_project_mfma $dirs | sed 's/^/project /'; } > $out/${tag}_crt_traffic.txt
rm -rf /tmp/pmc_crt_FETCH_SIZE /tmp/pmc_crt_WRITE_SIZE
# effective clock and power under the product kernels: GRBM_GUI_ACTIVE per launch (cycles, summed over the 8 XCDs) and
# rocm-smi sampled during a long run
for wl in B A C F; do
  rm -rf /tmp/pmc_clk_$wl
  (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_clk_$wl -- python $here/bench.py --workload $wl --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-traffic --no-rccl > /dev/null 2>&1)
  f=$(find /tmp/pmc_clk_$wl -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/${tag}_pmc_GRBM_GUI_ACTIVE_$wl.csv
  rm -rf /tmp/pmc_clk_$wl
done
smi() { for i in $(seq 1 $2); do rocm-smi -P -c --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done > $out/${tag}_smi_$1.jsonl; }
for wl in B C F E; do
  steps=$([ $wl = E ] && echo 60 || echo 3000)
  (smi $wl 36 &) ; timeout 120 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $out/${tag}_bench_${wl}_long.json 2>/dev/null; sleep 1.5
done
ls -la $out | tail -25
