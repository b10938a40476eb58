#!/bin/bash
# GPU session J of round 3 (the 8192-word rows on the persistent kernel; the operand A/B again with a CORRECT VGPR build): (1) the persistent, prefetching product of rows of 16384 words: parity, then same-box A/B against
# one workgroup per row (NFLHIP_XCD=0) on workload C and two other moduli counts; (2) a clean same-box A/B of SGPR vs VGPR
# operands in the metric kernel with checksums (session C's "old" library turned out not to be comparable).
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_xcd.py -x -q -k "persistent_rows" > $out/r03j_pytest.txt 2>&1
tail -5 $out/r03j_pytest.txt
: > $out/r03j_c_ab.txt
for rep in 1 2; do
  for x in 0 default; do
    if [ $x = default ]; then unset NFLHIP_XCD; else export NFLHIP_XCD=$x; fi
    r=$(timeout 300 python bench.py --workload G --steps 50 --warmup 5 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['kernel'])")
    echo "G rep$rep NFLHIP_XCD=$x: value frac kernel_ms kernel: $r" >> $out/r03j_c_ab.txt
  done
done
unset NFLHIP_XCD
cat $out/r03j_c_ab.txt
: > $out/r03j_operand_ab.txt
cp nfllib_amd/libnflhip.so /tmp/lib_shipped.so
for rep in 1 2 3; do
  for v in /tmp/lib_shipped.so build/vgpr_operands/nfllib_amd/libnflhip.so; do
    timeout 120 python tools/ab_probe.py $v 3 2>&1 | grep -v amdgpu.ids >> $out/r03j_operand_ab.txt
  done
done
(cd /tmp && for v in /tmp/lib_shipped.so $GRAFT_REPO_ROOT/build/vgpr_operands/nfllib_amd/libnflhip.so; do
  rm -rf /tmp/pmc_ab; timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_ab -- python $GRAFT_REPO_ROOT/tools/ab_probe.py $v 0.3 > /dev/null 2>&1
  f=$(find /tmp/pmc_ab -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$v" >> $GRAFT_REPO_ROOT/gpurun_out/r03j_operand_ab.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "polymul4096" in r["Kernel_Name"]]
c = [float(r["Counter_Value"]) / 8 for r in rows]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
print("%s: %d launches, GRBM_GUI_ACTIVE / 8 = %.0f cycles (min %.0f max %.0f), %.0f ns, %.0f MHz" % (sys.argv[2], len(rows), sum(c) / max(1, len(c)), min(c or [0]), max(c or [0]), sum(d) / max(1, len(d)), 1e3 * sum(c) / max(1, sum(d))))
PY
done)
cat $out/r03j_operand_ab.txt
