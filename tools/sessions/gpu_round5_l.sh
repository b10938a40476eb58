#!/bin/bash
# round 5, session l: the compiled 64-bit wave-per-row kernels at 2 / 3 / 4 workgroups per CU (-DNFLHIP_W64_OCC)
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
{
for lib in shipped occ3 occ4; do
  p=$here; [ $lib != shipped ] && p=$here/build/$lib
  for cfg in "1024 2 32768" "2048 2 16384" "1024 8 8192"; do
    set -- $cfg
    echo -n "$lib polymul n $1 nm $2: "; PYTHONPATH=$p python tools/probes/hold_polymul.py $1 $2 $3 1.0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['polymul_per_s'], d['frac'])"
  done
  for cfg in "1024 2 32768" "2048 2 16384"; do
    set -- $cfg
    PYTHONPATH=$p python tools/lwe_demo.py --limb-bits 64 --degree $1 --nmoduli $2 --batch $3 --plan fused --fixed-key --reps 10 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib lwe', d['degree'], 'enc/s', d['encryptions_per_s'], 'dec/s', d['decryptions_per_s'], d['decrypts_to_zero'])"
  done
done
} > gpurun_out/r05_w64_occ.txt 2>&1
cat gpurun_out/r05_w64_occ.txt
