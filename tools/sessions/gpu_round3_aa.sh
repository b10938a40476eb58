#!/bin/bash
# GPU session AA of round 3: L1 -> L2 read requests of the 32768-word kernels with the lane-major twiddle copy (shipped) and with
# the natural order (build/natural_tw), rocprofv3 --pmc in its own pass (workload F, 6 products).
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
here=$(pwd)
cp nfllib_amd/libnflhip.so /tmp/lib_lm.so
cp build/natural_tw/nfllib_amd/libnflhip.so /tmp/lib_nat.so
: > $out/r03aa_tcp.txt
for v in lm nat; do
  cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
  for ctr in TCP_TCC_READ_REQ_sum TCC_REQ_sum; do
    rm -rf /tmp/pmc_aa
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_aa -- python $here/bench.py --workload F --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-traffic --no-rccl > /dev/null 2>&1)
    f=$(find /tmp/pmc_aa -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$v" "$ctr" >> $out/r03aa_tcp.txt <<'PY'
import csv, sys, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "32768" in r["Kernel_Name"]:
        by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(by.items()):
    print("%-4s %-22s %-32s %2d launches  %.4g per launch" % (sys.argv[2], sys.argv[3], k, len(v), sum(v) / len(v)))
PY
  done
done
cp /tmp/lib_lm.so nfllib_amd/libnflhip.so
cat $out/r03aa_tcp.txt
