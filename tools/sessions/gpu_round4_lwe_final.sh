out=gpurun_out
rm -f $out/r04_final_lwe.jsonl
for shape in "64 4096 4 16384" "64 8192 2 8192" "64 16384 8 1024" "64 32768 2 1024" "64 1024 2 65536" "32 1024 2 65536"; do
  set -- $shape
  for plan in unfused fused; do
    python tools/lwe_demo.py --limb-bits $1 --degree $2 --nmoduli $3 --batch $4 --plan $plan --reps 10 --fixed-key 2>/dev/null >> $out/r04_final_lwe.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r04_final_lwe.jsonl'):
    d = json.loads(l); print(d['limb_bits'], d['degree'], d['nmoduli'], d['batch'], d['plan'], d['encryptions_per_s'], d['decryptions_per_s'], d['decrypts_to_zero'])
PY
