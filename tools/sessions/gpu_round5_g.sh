#!/bin/bash
# round 5, session g: sampler tests, LWE demo with both draws, C++ header programs
cd "$(dirname "$0")/../.."
here=$(pwd)
export PYTHONPATH=$here TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_samplers.py tests/test_gpu_fused.py tests/test_cpp_surface.py tests/test_zz_gpu_deferred_loops.py tests/test_reference_programs.py -m gpu -q 2>&1 | cut -c1-400 | tail -30 > gpurun_out/r05_g_tests.txt
tail -12 gpurun_out/r05_g_tests.txt
{
for db in 64 32; do for plan in unfused fused; do
  python tools/lwe_demo.py --degree 4096 --nmoduli 4 --batch 8192 --plan $plan --draw-bits $db --fixed-key --reps 10 2>/dev/null | grep '^{'
done; done
for db in 64 32; do python tools/lwe_demo.py --degree 16384 --nmoduli 8 --batch 1024 --plan fused --draw-bits $db --fixed-key --reps 10 2>/dev/null | grep '^{'; done
for db in 64 32; do python tools/lwe_demo.py --degree 1024 --nmoduli 2 --limb-bits 32 --batch 65536 --plan fused --draw-bits $db --fixed-key --reps 10 2>/dev/null | grep '^{'; done
NFL_LWE_REPS=16384 tests/cpp/resident_test | grep '^{'
NFL_HIP_WIDE_DRAWS=1 NFL_LWE_REPS=16384 tests/cpp/resident_test | grep '^{'
} > gpurun_out/r05_lwe_draws.txt 2>&1
cat gpurun_out/r05_lwe_draws.txt
