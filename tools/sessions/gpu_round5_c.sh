#!/bin/bash
# round 5, session c: workload E with the pipeline's scratch rows aliased onto 4 / 16 / 64 rows (NFL_GEN_ABLATE=scratchN, wrong
# results by construction): the forward pass's writes and the block products' reads are then served by the on-die caches.
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
{
for lib in shipped scratch4 scratch16 scratch64 shipped; do
  p=$here; [ $lib != shipped ] && p=$here/build/abl_$lib
  for b in 64 128; do
    echo -n "$lib batch $b: "; NFLHIP_XCD=0 PYTHONPATH=$p timeout 120 python tools/probes/hold_polymul.py 65536 30 $b 3
  done
done
} > gpurun_out/r05_E_scratch_alias.txt 2>&1
cat gpurun_out/r05_E_scratch_alias.txt
