#!/usr/bin/env python3
"""Per stream of a rocprofv3 kernel trace of bench.py: the launches of the last launch sequences in order, with each launch's
duration and the idle gap in front of it on its own stream (a dependent chain: a gap is time the chain did not run).
usage: tools/seq_gaps.py KERNEL_TRACE.csv [--last N] [--summary]"""
import csv
import sys
import collections

path = sys.argv[1]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 90
rows = [r for r in csv.DictReader(open(path)) if "rocclr" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Stream_Id"]].append(r)
for sid, v in sorted(by.items()):
    v = [r for r in v if any(k in r["Kernel_Name"] for k in ("gn_", "prep_", "stage_in", "ica_hinv"))]
    if len(v) < 40:
        continue
    tail = v[-last:]
    gaps = collections.defaultdict(list)
    prev = None
    pos = 0
    for r in tail:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ellc::", "")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "stage_in" in n:
            pos = 0
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        gaps[pos].append(gap)
        if "--summary" not in sys.argv:
            print("stream %s #%2d %-30s grid %6d x %3d  dur %7.1f us  gap %7.1f us" % (sid, pos, n[:30], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), (e - s) / 1e3, gap))
        prev = e
        pos += 1
    print("stream %s: mean gap by position in the sequence:" % sid, " ".join("%d:%.0f" % (p, sum(g) / len(g)) for p, g in sorted(gaps.items()) if sum(g) / len(g) > 2.0))
