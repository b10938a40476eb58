#!/bin/bash
# round 5, session e: workload F (u64/32768/2) with the scratch operand b' of its composed product aliased onto 4 / 64 row blocks
# (NFL_GEN_ABLATE=bprimeN, wrong results by construction): what is the prize of a plan whose b' never reaches HBM?
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
{
for lib in shipped bprime4 bprime64 shipped; do
  p=$here; [ $lib != shipped ] && p=$here/build/abl_$lib
  for b in 512 2048; do
    echo -n "$lib batch $b: "; NFLHIP_XCD=0 PYTHONPATH=$p timeout 120 python tools/probes/hold_polymul.py 32768 2 $b 3 2>/dev/null
  done
done
export TMPDIR=/tmp
for lib in shipped bprime4; do
  p=$here; [ $lib != shipped ] && p=$here/build/abl_$lib
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_f; (cd /tmp && NFLHIP_XCD=0 PYTHONPATH=$p rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_f -- python $here/tools/probes/hold_polymul.py 32768 2 512 0.05 > /dev/null 2>&1)
    echo -n "$lib $c (KiB summed over the product's dispatches, FETCH x2 on gfx950): "; python - <<PY
import csv, glob
tot = {}
for f in glob.glob("/tmp/pmc_f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "$c" and "nflhip_" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += float(r["Counter_Value"])
print({k: (v[0], round(v[1] / v[0], 1)) for k, v in tot.items()})
PY
  done
done
} > gpurun_out/r05_F_bprime_alias.txt 2>&1
cat gpurun_out/r05_F_bprime_alias.txt
