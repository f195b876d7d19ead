#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py with several batches in flight: how much of the wall time has at least one
kernel running, how many run at once on average, and the time share per kernel family (steady-state window only).
usage: overlap_summary.py <trace dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:24], r.get("Queue_Id", "?"),
             int(r.get("Grid_Size_Y", 1) or 1)) for r in rows)
# steady-state window: from the 10th to the 50th batch of the pipelined run (compaction launches with grid.y = batch size)
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 32
preps = [e[0] for e in ev if e[2].startswith("prep_count") and e[4] == BATCH]
if len(preps) < 55:
    sys.exit("not enough batches in the trace")
t0, t1 = preps[10], preps[50]
nb = sum(1 for p in preps if t0 <= p < t1)
win = [(max(s, t0), min(e, t1), n, q) for s, e, n, q, _ in ev if e > t0 and s < t1]
pts = sorted([(s, 1) for s, e, _, _ in win] + [(e, -1) for s, e, _, _ in win])
busy = 0
area = 0
depth = 0
last = t0
for t, d in pts:
    if depth > 0:
        busy += t - last
    area += depth * (t - last)
    depth += d
    last = t
span = t1 - t0
print("window %.1f us, %d batches -> %.1f us per batch" % (span / 1e3, nb, span / 1e3 / nb))
print("some kernel running %.1f %% of the time; mean kernels in flight %.2f; queues used: %s" % (100.0 * busy / span, area / span, sorted(set(q for _, _, _, q in win))))
fam = collections.Counter()
for s, e, n, _ in win:
    fam[n] += e - s
for n, v in fam.most_common(8):
    print("  %-26s %7.1f us per batch (sum of durations)" % (n, v / 1e3 / nb))
