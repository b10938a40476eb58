#!/bin/bash
# round 6, session l: role stamps of workload E on the one-launch plan (s_memtime per role: workgroup free / inputs ready / done):
# where do the 768 persistent workgroups spend their time?  batch 128 and 256, block products on incomplete / complete transforms
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for cfgs in "65536 30 128 2" "65536 30 128 0" "65536 30 256 2" "32768 2 2048 2"; do
  PYTHONPATH=$(pwd) NFLHIP_XCD=1 timeout 300 python tools/xcd_trace.py $cfgs 2>/dev/null
  echo
done
} > gpurun_out/r06_E_role_stamps.txt 2>&1
cat gpurun_out/r06_E_role_stamps.txt
