#!/usr/bin/env python3
"""Per-(kernel, grid) median durations and the per-step composition from a rocprofv3 kernel trace CSV."""
import collections
import csv
import glob
import statistics
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows:
    d[(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ellc::", "")[:44], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    if len(v) >= 5 and ("gn_" in k[0] or "prep" in k[0]):
        print("%-44s grid=(%6s,%3s) n=%4d median %7.2f us  total %9.1f us" % (k[0], k[1], k[2], len(v), statistics.median(v), sum(v)))
idx = [i for i, r in enumerate(rows) if "prep_count" in r["Kernel_Name"] and r["Grid_Size_Y"] == "32"]
if len(idx) > 3:
    a, b = idx[-3], idx[-2]
    seg = rows[a:b]
    last = max(i for i, r in enumerate(seg) if "gn_" in r["Kernel_Name"])
    seg = seg[: last + 1]
    print("one B=32 step: %d kernels, span %.1f us" % (len(seg), (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3))
