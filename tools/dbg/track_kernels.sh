#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06}/track_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_track.py 120 fast ${2:-} > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
anchor = "stage_in_args" if sum("stage_in_args" in n for n in names) > 40 else "gn_fca_persist"   # r06: the staging launch is folded into the resident one
starts = [i for i, n in enumerate(names) if anchor in n]
i0, i1 = starts[-20], starts[-19]
t0 = int(rows[i0]["Start_Timestamp"])
prev = None
for r in rows[i0 - 3:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ellc::", "")[:34]
    print("%-34s start %7.1f dur %6.1f gap %6.1f  grid %s x %s" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"]))
    prev = e
PY
