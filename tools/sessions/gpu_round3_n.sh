#!/bin/bash
# GPU session N of round 3: (1) does the operand POSITION in v_mad_u64_u32 matter (multiplicand / multiplier exchanged in every
# butterfly multiply-add: build/swap_mad, NFL_GEN_SWAP_MAD=1)?  metric kernel, same box, checksums, power; (2) bench.py's new
# in-run power sample on the default line.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cp nfllib_amd/libnflhip.so /tmp/lib_shipped.so
smi() { for i in $(seq 1 $2); do rocm-smi -P -c --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done > $out/r03n_smi_$1.jsonl; }
: > $out/r03n_swap_ab.txt
for rep in 1 2 3; do
  for v in shipped swap_mad; do
    lib=/tmp/lib_shipped.so; [ $v = swap_mad ] && lib=build/swap_mad/nfllib_amd/libnflhip.so
    if [ $rep = 1 ]; then (smi $v 24 &); sleep 0.5; fi
    timeout 120 python tools/ab_probe.py $lib 4 2>&1 | grep -v amdgpu.ids >> $out/r03n_swap_ab.txt
    sleep 1.5
  done
done
cat $out/r03n_swap_ab.txt
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r03n_smi_*.jsonl")):
    rows = []
    for line in open(f):
        try:
            d = json.loads(line)["card0"]
        except Exception:
            continue
        p = next((float(v) for k, v in d.items() if "Power" in k), None)
        c = next((v for k, v in d.items() if k.startswith("sclk")), "")
        rows.append((p, int("".join(ch for ch in c.split("(")[-1] if ch.isdigit()) or 0)))
    busy = [r for r in rows if r[0] and r[0] > 600][1:-1]
    if busy:
        print("%-32s %2d busy samples: %.0f W, sclk %.0f MHz" % (os.path.basename(f), len(busy), sum(r[0] for r in busy) / len(busy), sum(r[1] for r in busy) / len(busy)))
PY
timeout 600 python bench.py > $out/r03n_bench_default.json 2> $out/r03n_bench_default.err
python -c "import json; d=json.loads(open('gpurun_out/r03n_bench_default.json').readline()); print(d['value'], d['roofline']['frac'], d['roofline']['secondary'])"
