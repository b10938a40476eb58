#!/bin/bash
# round 6, session z: the one-chain multiply-add of the 64-bit transform-fused kernels (tools/asmgen/fused.py mac128: x y + addend as ONE
# 128-bit chain, 27 / 23-25 instructions per word where product - fold - add - fold took 33 / 26-28): parity (fused + fuzz + deferred loops),
# then the LWE demo against the previous library (build/old_fma), alternated, at the reference's demo degrees and the metric's shape
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fuzz.py tests/test_zz_gpu_deferred_loops.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/r06_mac128_tests.txt
cat gpurun_out/r06_mac128_tests.txt
cp nfllib_amd/libnflhip.so /tmp/lib_new.so
{
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp build/old_fma/nfllib_amd/libnflhip.so nfllib_amd/libnflhip.so; else cp /tmp/lib_new.so nfllib_amd/libnflhip.so; fi
    for cfgs in "4096 4 8192" "8192 2 8192" "16384 2 4096" "32768 2 2048" "1024 2 32768" "2048 2 16384"; do
      set -- $cfgs
      PYTHONPATH=$here python tools/lwe_demo.py --limb-bits 64 --degree $1 --nmoduli $2 --batch $3 --plan fused --fixed-key --reps 10 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v lwe u64 $1 $2 enc/s', d['encryptions_per_s'], 'dec/s', d['decryptions_per_s'], d['decrypts_to_zero'], d['digest'])"
    done
  done
done
} > gpurun_out/r06_mac128_ab.txt 2>&1
cp /tmp/lib_new.so nfllib_amd/libnflhip.so
cat gpurun_out/r06_mac128_ab.txt
