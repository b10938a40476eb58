#!/usr/bin/env python3
"""Mean per-launch value of SQ counters for the product kernel from rocprofv3 --pmc passes (one directory per pass).
    python tools/pmc_sq.py <kernel substring> dir1 dir2 ..."""
import csv, glob, os, sys
sub, dirs = sys.argv[1], sys.argv[2:]
acc = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                e = acc.setdefault(r["Counter_Name"], [0, 0.0])
                e[0] += 1; e[1] += float(r["Counter_Value"])
for k in sorted(acc):
    print("%-24s mean_per_launch=%g n=%d" % (k, acc[k][1] / acc[k][0], acc[k][0]))
