#!/bin/bash
# round 6, session ag: Gaussian tables computed once per process and primed by FastGaussianNoise's constructor (where the reference builds its
# MPFR table): sampler / surface tests, then the reference's own timing programs side by side
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/test_gpu_samplers.py tests/test_reference_programs.py tests/test_cpp_surface.py tests/test_zz_gpu_deferred_loops.py tests/test_gpu_fused.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -4 > gpurun_out/r06_gauss_cache_tests.txt
cat gpurun_out/r06_gauss_cache_tests.txt
timeout 1500 python tools/reference_programs.py --reps 3 --json gpurun_out/r06_reference_programs.json > gpurun_out/r06_reference_programs.txt 2> gpurun_out/r06_reference_programs.err
grep -n "gaussian\|LWE" gpurun_out/r06_reference_programs.txt | cut -c1-150; tail -2 gpurun_out/r06_reference_programs.txt; tail -3 gpurun_out/r06_reference_programs.err
