#!/usr/bin/env python3
"""Median duration of the fused Gauss-Newton launches per grid (i.e. per level) in a rocprofv3 kernel trace of bench.py:
usage: level_durations.py <trace dir>"""
import collections
import csv
import glob
import statistics
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].split("::")[-1]
    if not any(t in n for t in ("gn_", "prep_", "stage_in", "ica_hinv")):
        continue
    key = (n[:28], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]))
    by[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s blocks (%5d, %4d)  n=%5d  median %8.2f us  total %9.1f us" % (k[0], k[1], k[2], len(v), statistics.median(v), sum(v)))
