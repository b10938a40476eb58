#!/bin/bash
# round 6, session v: the round-3 power ablation repeated on the CURRENT metric kernel (nflhip_polymul4096i2_asm, incomplete
# transforms): shipped / operands in the L2 (row0) / no HBM + twiddle + LDS traffic (tw0,nolds,row0) / no butterflies, each held 6 s
# with rocm-smi sampled beside it.  bench.py's ceiling_frac_no_memory quoted the round-3 kernel until now.
cd "$(dirname "$0")/../.."
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
smi() { for i in $(seq 1 $2); do rocm-smi -P -c --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done > $out/r06v_smi_$1.jsonl; }
cp nfllib_amd/libnflhip.so /tmp/lib_shipped.so
: > $out/r06v_ablate.txt
for v in shipped tw0_nolds_row0 row0 nobfly shipped2; do
  case $v in shipped*) cp /tmp/lib_shipped.so nfllib_amd/libnflhip.so;; *) cp build/abl_$v/nfllib_amd/libnflhip.so nfllib_amd/libnflhip.so;; esac
  (smi B_$v 44 &)
  sleep 1
  echo "== $v" >> $out/r06v_ablate.txt
  timeout 120 python tools/power_probe.py B 6 >> $out/r06v_ablate.txt 2>&1
  sleep 2
done
cp /tmp/lib_shipped.so nfllib_amd/libnflhip.so
python - <<'PY' > gpurun_out/r06v_summary.txt
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r06v_smi_*.jsonl")):
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
PY
cat $out/r06v_summary.txt $out/r06v_ablate.txt
