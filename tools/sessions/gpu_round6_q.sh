#!/bin/bash
# round 6, session q: the device's timeline under the poly_p LWE loop (rocprofv3 --kernel-trace): where does it idle?
cd "$(dirname "$0")/../.."
R=$(pwd)
mkdir -p gpurun_out
make -s -C tests/cpp resident_test 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for t in 0 1; do
  rm -rf /tmp/kt$t
  NFL_HIP_QUEUE_MIN=1024 NFL_HIP_QUEUE_THREAD=$t NFL_LWE_REPS=16384 rocprofv3 --kernel-trace -d /tmp/kt$t -o kt --output-format csv -- $R/tests/cpp/resident_test > /dev/null 2>&1
  f=$(find /tmp/kt$t -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $t <<'PY' > $R/gpurun_out/r06_lwe_timeline_t$t.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows)
print("# thread", sys.argv[2], "kernels", len(ev))
# find the fused kernels' launches and print a window: name, start (us since first), duration, gap before
t0 = ev[0][0]
fused = [i for i, e in enumerate(ev) if "fwd_fma" in e[2] or "fused" in e[2].lower()]
print("# fused launches:", len(fused))
prev_end = None
for i, (s, e, n) in enumerate(ev):
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = max(prev_end or 0, e)
    print("%10.1f us  dur %8.1f us  gap %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
PY
done
cd $R
for t in 0 1; do echo "== thread $t"; grep -n "fused_enc2" gpurun_out/r06_lwe_timeline_t$t.txt | awk '{print $1,$2,$5}' | sed -n 1,60p | tr '\n' ';'; echo; done
