#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
export PYTHONPATH=$(pwd)
for b in 32 64 128 192; do NFLHIP_XCD=0 timeout 200 python tools/probes/two_stream_probe.py 65536 30 $b 30 2>&1 | tail -2; done
for b in 256 1024; do NFLHIP_XCD=0 timeout 200 python tools/probes/two_stream_probe.py 32768 2 $b 100 2>&1 | tail -2; done
for b in 1024; do timeout 200 python tools/probes/two_stream_probe.py 16384 8 $b 100 2>&1 | tail -2; done
} > gpurun_out/r05_two_streams.txt 2>&1
cat gpurun_out/r05_two_streams.txt
