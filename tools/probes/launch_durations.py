#!/usr/bin/env python3
"""Per-dispatch durations out of a rocprofv3 --kernel-trace directory, in launch order, for kernels whose name contains argv[2]:
the pipeline plans are several launches per product with different role mixes -- the average hides which of them is slow."""
import csv
import glob
import os
import sys

d, pat = sys.argv[1], sys.argv[2]
period = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Grid_Size_X") or r.get("Grid_Size") or "?"))
rows.sort()
print("%d dispatches of *%s*" % (len(rows), pat))
tail = rows[-3 * period:]
for i, (s, e, n, g) in enumerate(tail):
    gap = (s - tail[i - 1][1]) / 1e3 if i else 0.0
    print("%3d %-44s grid %8s  %9.1f us   gap before %7.1f us" % (i % period, n, g, (e - s) / 1e3, gap))
if len(rows) >= period:
    span = (rows[-1][1] - rows[-period][0]) / 1e3
    busy = sum(e - s for s, e, _, _ in rows[-period:]) / 1e3
    print("last product: %.1f us wall, %.1f us inside kernels" % (span, busy))
