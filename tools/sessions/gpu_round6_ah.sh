#!/bin/bash
# round 6, session ah: the comparison flags (expr::operator bool, CHECK_STRICTMOD) in pinned host memory with a per-call token instead of a
# clearing pass + a copy back: parity of everything that compares, the per-call probe
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/test_abi.py tests/test_gpu_parity.py tests/test_cpp_surface.py tests/test_reference_programs.py tests/test_zz_gpu_deferred_loops.py tests/test_gpu_fuzz.py tests/test_gpu_big_delta.py tests/test_gpu_graph.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -4 > gpurun_out/r06_flag_tests.txt
cat gpurun_out/r06_flag_tests.txt
{
python tools/probes/first_use.py 64 8192 2 2>&1 | grep -v amdgpu.ids
python tools/probes/first_use.py 32 1024 2 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_flag_probe.txt 2>&1
grep "a == b" gpurun_out/r06_flag_probe.txt
