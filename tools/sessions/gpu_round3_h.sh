#!/bin/bash
# GPU session H of round 3: how much of the long-row kernels' time is exchange / barrier / memory phase?  C (n = 16384) and
# F (n = 32768) with the LDS exchanges dropped, with the barriers dropped as well, with the rows in the L2, and with the
# butterflies dropped (tools/sessions/build_ablations.sh).  Plus a same-box A/B of the metric kernel against round 2's library.
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cp nfllib_amd/libnflhip.so /tmp/lib_shipped.so
: > $out/r03h_ablate.txt
for v in shipped nolds nolds_nobar row0 nobfly; do
  if [ $v = shipped ]; then cp /tmp/lib_shipped.so nfllib_amd/libnflhip.so; else cp build/abl_$v/nfllib_amd/libnflhip.so nfllib_amd/libnflhip.so; fi
  for wl in C F G; do
    echo "== $v $wl" >> $out/r03h_ablate.txt
    timeout 120 python tools/power_probe.py $wl 2 2>&1 | grep -v amdgpu.ids >> $out/r03h_ablate.txt
  done
done
cp /tmp/lib_shipped.so nfllib_amd/libnflhip.so
: > $out/r03h_r2_ab.txt
for rep in 1 2 3; do
  for v in /tmp/lib_shipped.so build/r2/nfllib_amd/libnflhip.so; do
    timeout 120 python tools/ab_probe.py $v 3 2>&1 | grep -v amdgpu.ids >> $out/r03h_r2_ab.txt
  done
done
(cd /tmp && for v in /tmp/lib_shipped.so $GRAFT_REPO_ROOT/build/r2/nfllib_amd/libnflhip.so; do
  rm -rf /tmp/pmc_ab; timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_ab -- python $GRAFT_REPO_ROOT/tools/ab_probe.py $v 0.3 > /dev/null 2>&1
  f=$(find /tmp/pmc_ab -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$v" >> $GRAFT_REPO_ROOT/gpurun_out/r03h_r2_ab.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "polymul4096" in r["Kernel_Name"]]
c = [float(r["Counter_Value"]) / 8 for r in rows]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
print("%s: %d launches of %s, GRBM_GUI_ACTIVE / 8 = %.0f cycles (min %.0f max %.0f), %.0f ns, %.0f MHz" % (sys.argv[2], len(rows), rows[0]["Kernel_Name"] if rows else "-", sum(c) / max(1, len(c)), min(c or [0]), max(c or [0]), sum(d) / max(1, len(d)), 1e3 * sum(c) / max(1, sum(d))))
PY
done)
cp /tmp/lib_shipped.so nfllib_amd/libnflhip.so
grep -E "==|last third" $out/r03h_ablate.txt; cat $out/r03h_r2_ab.txt
