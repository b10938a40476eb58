#!/usr/bin/env python3
"""HOST-SIDE PROFILING AID: aggregates `addr2line -a -f -i` output of the PC samples (pcsample.h) by the innermost frame
that lies in nfl.hpp (or by function for library code).  Usage: pcsample_report.py /tmp/a2l.txt [rows]"""
import collections
import sys

outer = collections.Counter()
frames = []


def flush():
    if frames:
        for fn, loc in frames:
            if "nfl_hip/" in loc:
                outer[loc.split("/")[-1].split(" ")[0]] += 1
                break
        else:
            outer[frames[0][0][:70] + " @ " + frames[0][1].split("/")[-1]] += 1


lines = open(sys.argv[1]).read().split("\n")
i = 0
while i < len(lines):
    if lines[i].startswith("0x"):
        flush()
        frames = []
        i += 1
    elif i + 1 < len(lines) and not lines[i + 1].startswith("0x"):
        frames.append((lines[i], lines[i + 1]))
        i += 2
    else:
        i += 1
flush()
tot = sum(outer.values())
print("samples", tot)
for k, v in outer.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    print("%5.1f%%  %s" % (100.0 * v / tot, k))
