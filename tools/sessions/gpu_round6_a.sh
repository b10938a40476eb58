#!/bin/bash
# round 6, session a: state of the tree on this round's box (gpu tests, driver-style bench line) and the fetch-width ablation
# of the metric kernel (NFL_GEN_ABLATE=x4: 8 x 16-byte coefficient fetches / stores per thread instead of 16 x 8-byte ones;
# wrong results by construction, same instruction stream otherwise), alternated with the shipped library on the same box.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for rep in 1 2 3; do
  for lib in nfllib_amd/libnflhip.so build/abl_x4/nfllib_amd/libnflhip.so; do
    python tools/ab_probe.py $lib 3 2>&1 | tail -1
  done
done
} > gpurun_out/r06_fetch_width_ab.txt 2>&1
cat gpurun_out/r06_fetch_width_ab.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_start.json 2> gpurun_out/r06_bench_start.err
tail -c 600 gpurun_out/r06_bench_start.json
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06_gputests_start.txt
cat gpurun_out/r06_gputests_start.txt
