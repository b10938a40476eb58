#!/bin/bash
# GPU session G of round 3: where do the 1 400 W go?  (1) the pure-arithmetic streams of tools/ubench_issue.hip held for
# seconds (session C's launches were 1 - 6 ms: too short for the power controller); (2) the metric kernel with one
# ingredient removed at a time (tools/sessions/build_ablations.sh), each held for 6 s: rate, power, clock.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
smi() { for i in $(seq 1 $2); do rocm-smi -P -c --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done > $out/r03g_smi_$1.jsonl; }
: > $out/r03g_sustain.txt
for st in bfly64iv_4 op_mad64_4 bfly64_mads_only bfly64_light_only op_add_4; do
  (smi ub_$st 34 &)
  sleep 1
  timeout 60 ./build/ubench_issue --sustain $st 8 6 >> $out/r03g_sustain.txt 2>&1
  sleep 2
done
cp nfllib_amd/libnflhip.so /tmp/lib_shipped.so
: > $out/r03g_ablate.txt
for v in shipped tw0 nolds row0 nobfly tw0_nolds_row0; do
  if [ $v = shipped ]; then cp /tmp/lib_shipped.so nfllib_amd/libnflhip.so; else cp build/abl_$v/nfllib_amd/libnflhip.so nfllib_amd/libnflhip.so; fi
  (smi B_$v 44 &)
  sleep 1
  echo "== $v" >> $out/r03g_ablate.txt
  timeout 120 python tools/power_probe.py B 6 >> $out/r03g_ablate.txt 2>&1
  sleep 2
done
cp /tmp/lib_shipped.so nfllib_amd/libnflhip.so
python - <<'PY' > gpurun_out/r03g_summary.txt
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r03g_smi_*.jsonl")):
    rows = []
    for line in open(f):
        try:
            d = json.loads(line)["card0"]
        except Exception:
            continue
        p = next((float(v) for k, v in d.items() if "Power" in k), None)
        c = next((v for k, v in d.items() if k.startswith("sclk")), "")
        mhz = int("".join(ch for ch in c.split("(")[-1] if ch.isdigit()) or 0)
        rows.append((p, mhz))
    busy = [r for r in rows if r[0] and r[0] > 600]
    if busy:
        busy = busy[1:-1] or busy
        print("%-40s %2d busy samples: %.0f W mean (max %.0f), sclk %.0f MHz mean (min %d)" % (os.path.basename(f), len(busy), sum(r[0] for r in busy) / len(busy), max(r[0] for r in busy), sum(r[1] for r in busy) / len(busy), min(r[1] for r in busy)))
    else:
        print("%-40s no busy sample of %d; max %.0f W" % (os.path.basename(f), len(rows), max([r[0] or 0 for r in rows] or [0])))
PY
cat $out/r03g_summary.txt; cat $out/r03g_ablate.txt; grep -E "t=(1.0|3.0|5.5|6.0)" $out/r03g_sustain.txt
