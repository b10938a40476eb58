#!/bin/bash
# round 6, session al: host-pointer expression trees with four distinct operands (c = c + shoup(a * b, b') in one call): parity, the
# reference's programs once more
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_cpp_surface.py tests/test_reference_programs.py tests/test_zz_gpu_deferred_loops.py tests/test_gpu_fuzz.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -4 > gpurun_out/r06_eval4_tests.txt
cat gpurun_out/r06_eval4_tests.txt
timeout 1500 python tools/reference_programs.py --reps 3 --json gpurun_out/r06_reference_programs.json > gpurun_out/r06_reference_programs.txt 2> gpurun_out/r06_reference_programs.err
grep -n "FMA" gpurun_out/r06_reference_programs.txt | cut -c1-150; tail -1 gpurun_out/r06_reference_programs.txt | cut -c1-120; tail -3 gpurun_out/r06_reference_programs.err
