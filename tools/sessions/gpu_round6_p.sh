#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
make -s -C tests/cpp resident_test 2>&1 | tail -3
{
for reps in 16384 2048; do
for t in 0 1; do
echo "## thread $t, reps $reps"
NFL_LWE_VERBOSE=1 NFL_HIP_QUEUE_STATS=1 NFL_HIP_QUEUE_THREAD=$t NFL_LWE_REPS=$reps tests/cpp/resident_test 2>&1 >/dev/null | grep -v "queue run: [0-9]* rec" | grep "lwe:\|collects: waited [1-9]" | head -3
done
done
echo "## GPU side: the fused batch entry points at the run lengths of the queue (encryptions per launch)"
PYTHONPATH=$(pwd) python - <<'PY'
import torch, time
from nfllib_amd import Engine
import tools.lwe_demo as L
print("lwe_demo", [n for n in dir(L) if not n.startswith("_")][:40])
PY
} > gpurun_out/r06_queue_thread3.txt 2>&1
cat gpurun_out/r06_queue_thread3.txt
