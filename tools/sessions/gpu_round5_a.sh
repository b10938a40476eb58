#!/bin/bash
# round 5, session a: the Infinity Cache probe and workload E with every chunk aliased into a cache-sized window
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== mall_probe"; timeout 300 build/mall_probe
echo "== E alias ablation (library build/abl, -DNFLHIP_ABLATION_KNOBS)"
PYTHONPATH=build/abl timeout 900 python tools/probes/e_alias_probe.py 64
} > gpurun_out/r05_mall_E.txt 2>&1
tail -60 gpurun_out/r05_mall_E.txt
