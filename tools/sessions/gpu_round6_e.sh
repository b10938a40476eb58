#!/bin/bash
# round 6, session e: (1) contexts past the 92nd 62-bit modulus with the per-row family split (rows 0-91 on the generated delta-form
# kernels, the rest on the general-modulus kernels): parity + rates at 92 / 93 / 96 moduli; (2) the reference's own timing programs,
# real NFLlib on the host CPU vs the drop-in header on the GPU; (3) fuzz with the tree's seed; (4) the driver's bench command
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_big_delta.py tests/test_gpu_fuzz.py -x -q 2>&1 | grep -E "passed|failed|rror|FUZZ" | tail -6 > gpurun_out/r06_big_delta_tests.txt
cat gpurun_out/r06_big_delta_tests.txt
{
for nm in 92 93 96; do
  echo -n "polymul n 4096 nm $nm batch 512: "; PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py 4096 $nm 512 2 2>/dev/null
done
} > gpurun_out/r06_big_delta_rates.txt 2>&1
cat gpurun_out/r06_big_delta_rates.txt
timeout 1500 python tools/reference_programs.py --reps 3 --json gpurun_out/r06_reference_programs.json > gpurun_out/r06_reference_programs.txt 2> gpurun_out/r06_reference_programs.err
tail -5 gpurun_out/r06_reference_programs.txt; tail -3 gpurun_out/r06_reference_programs.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_e.json 2> gpurun_out/r06_bench_e.err
tail -c 400 gpurun_out/r06_bench_e.json
