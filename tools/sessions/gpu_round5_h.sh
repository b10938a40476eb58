#!/bin/bash
# round 5, session h: the short poly_p loop (2 048 encryptions): where its time goes, by queue limit
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for reps in 2048 4096; do
for lim in 1024 2048 3072 4096 8192; do
  echo "== reps $reps limit $lim"
  NFL_LWE_REPS=$reps NFL_LWE_VERBOSE=1 NFL_HIP_QUEUE_LIMIT=$lim tests/cpp/resident_test 2>&1 | grep "lwe:\|poly_p_enc" | cut -c1-260
done
echo "== reps $reps limit 8192 early run"
NFL_LWE_REPS=$reps NFL_LWE_VERBOSE=1 NFL_HIP_EARLY_RUN=1 tests/cpp/resident_test 2>&1 | grep "lwe:\|poly_p_enc" | cut -c1-260
done
} > gpurun_out/r05_short_loops.txt 2>&1
cat gpurun_out/r05_short_loops.txt
