#!/bin/bash
# round 6, session af: the host-pointer entry points on PINNED HOST staging buffers the kernels read and write directly (one polynomial per
# call, up to 1 MiB per operand; api.hip ensure_stage / Staged): the per-call probe, the tests of the host-pointer paths, the reference's own
# timing programs side by side once more
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
python tools/probes/first_use.py 64 8192 2 2>&1 | grep -v amdgpu.ids
python tools/probes/first_use.py 32 1024 2 2>&1 | grep -v amdgpu.ids
python tools/probes/first_use.py 64 32768 2 2>&1 | grep -v amdgpu.ids
python tools/probes/first_use.py 64 4096 4 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_host_calls_zero_copy.txt 2>&1
cat gpurun_out/r06_host_calls_zero_copy.txt
timeout 2000 python -m pytest tests/test_abi.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_reference_programs.py tests/test_cpp_surface.py tests/test_gpu_big_delta.py tests/test_gpu_samplers.py tests/test_gpu_u16_asm.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -4 > gpurun_out/r06_zero_copy_tests.txt
cat gpurun_out/r06_zero_copy_tests.txt
timeout 1500 python tools/reference_programs.py --reps 3 --json gpurun_out/r06_reference_programs.json > gpurun_out/r06_reference_programs.txt 2> gpurun_out/r06_reference_programs.err
tail -3 gpurun_out/r06_reference_programs.txt; tail -3 gpurun_out/r06_reference_programs.err
