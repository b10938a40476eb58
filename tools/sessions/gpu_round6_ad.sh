#!/bin/bash
# round 6, session ad: the one-time first-use costs paid at the first context of a device (api.hip warm_up_device): the probe before / after
# (NFLHIP_NO_WARMUP=1 = the lazy behaviour: a switch of that build, since removed), parity of what could be affected, then the reference's own timing programs side by side again
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== NFLHIP_NO_WARMUP=1 (lazy: every cost falls on the first call that needs it)"
NFLHIP_NO_WARMUP=1 python tools/probes/first_use.py 64 8192 2 2>&1 | grep -v amdgpu.ids
echo "== default (warm-up at the first context of the device)"
python tools/probes/first_use.py 64 8192 2 2>&1 | grep -v amdgpu.ids
python tools/probes/first_use.py 32 1024 2 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r06_first_use.txt 2>&1
cat gpurun_out/r06_first_use.txt
timeout 1500 python -m pytest tests/test_abi.py tests/test_gpu_parity.py tests/test_reference_programs.py tests/test_cpp_surface.py tests/test_gpu_graph.py tests/test_gpu_comm.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -4 > gpurun_out/r06_warmup_tests.txt
cat gpurun_out/r06_warmup_tests.txt
timeout 1500 python tools/reference_programs.py --reps 3 --json gpurun_out/r06_reference_programs.json > gpurun_out/r06_reference_programs.txt 2> gpurun_out/r06_reference_programs.err
tail -5 gpurun_out/r06_reference_programs.txt; tail -3 gpurun_out/r06_reference_programs.err
