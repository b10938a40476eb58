#!/bin/bash
# round 6, session k: determinism soak of every product plan (now on incomplete transforms), the generated wave-per-row kernels and their
# fused pipelines: 200 repetitions per shape, every result equal to the first
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
PYTHONPATH=$(pwd) timeout 1500 python tools/soak.py 200 > gpurun_out/r06_soak.txt 2>&1
tail -22 gpurun_out/r06_soak.txt
