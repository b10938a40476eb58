#!/bin/bash
# round 6, session g: workload E (u64/65536/30, batch 128 / 256), the chunked pipeline with SHORT first / last chunks: the first launch
# (forward role alone) and the last (inverse alone) are the pipeline's fill and drain.  Experiment build (-DNFLHIP_ABLATION_KNOBS):
# NFLHIP_PIPE_CHUNKS_RT = chunks, NFLHIP_PIPE_EDGE_RT = polynomials in the first and in the last chunk (0 = uniform)
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
{
for b in 128 256; do
  for cfgs in "4 0" "4 8" "4 16" "5 8" "5 16" "6 8" "6 16" "6 4" "8 8" "4 0"; do
    set -- $cfgs
    echo -n "batch $b chunks $1 edge $2: "; NFLHIP_XCD=0 NFLHIP_PIPE_CHUNKS_RT=$1 NFLHIP_PIPE_EDGE_RT=$2 PYTHONPATH=$here/build/knobs timeout 200 python tools/probes/hold_polymul.py 65536 30 $b 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['polymul_per_s'], d['frac'], d['package_W'], d['sclk_MHz'])"
  done
done
} > gpurun_out/r06_E_edge_chunks.txt 2>&1
cat gpurun_out/r06_E_edge_chunks.txt
