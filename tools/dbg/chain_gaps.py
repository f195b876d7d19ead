#!/usr/bin/env python3
"""Per hardware queue of a kernel trace of the pipelined bench: the launch sequences (stage_in ... gn_fused_finish), their span, the sum
of their kernels' durations and the gaps between consecutive kernels of a sequence; and how many kernels run at once.
usage: chain_gaps.py <trace dir>"""
import collections, csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
byq = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ellc::", "")
    byq[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r["Grid_Size_Y"] or 1)))
seqs = []
for q, ev in byq.items():
    ev.sort()
    cur = None
    for s, e, n, gy in ev:
        if n.startswith("stage_in"):
            cur = [(s, e, n)]
        elif cur is not None:
            cur.append((s, e, n))
            if n.startswith("gn_fused_finish"):
                if gy >= 1 and len(cur) > 30:
                    seqs.append((q, cur))
                cur = None
full = [c for q, c in seqs if len(c) in (34, 35, 36)]
print("sequences:", len(seqs), "full:", len(full))
full = full[len(full) // 4:]   # steady state
spans = [(c[-1][1] - c[0][0]) / 1e3 for c in full]
durs = [sum(e - s for s, e, _ in c) / 1e3 for c in full]
gaps = [[(c[i + 1][0] - c[i][1]) / 1e3 for i in range(len(c) - 1)] for c in full]
print("span median %.1f us, sum of kernel durations %.1f us, sum of gaps %.1f us (median gap %.2f us, max %.1f)" % (
    statistics.median(spans), statistics.median(durs), statistics.median(sum(g) for g in gaps), statistics.median(x for g in gaps for x in g), max(x for g in gaps for x in g)))
# per position
L = len(full[0])
for i in range(L):
    ds = [(c[i][1] - c[i][0]) / 1e3 for c in full if len(c) == L]
    gs = [(c[i][0] - c[i - 1][1]) / 1e3 for c in full if len(c) == L] if i else [0]
    print("%2d %-46s dur %7.1f  gap before %6.1f" % (i, full[0][i][2][:46], statistics.median(ds), statistics.median(gs)))
