#!/bin/bash
# Kernel timeline of the tracked-frame loop (fused call): rocprofv3 --kernel-trace of tools/dbg/track_modes.py, then the kernels of
# one steady-state frame with their start offsets and durations.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/track_trace3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ELLC_TRACK_NO_LC=1   # the two-thread loop crashes rocprofv3 (SIGSEGV inside its HIP interception; fine without the profiler)
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/dbg/track_modes.py > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# frames of the fused loop start with stage_in_args (which also counts the valid hypotheses)
starts = [i for i, n in enumerate(names) if "stage_in_args" in n]
import os as _os
k = int(_os.environ.get("ELLC_TRACE_FRAME", "30"))   # 30: the fused loop; 100: the loop of separate calls
i0, i1 = starts[k], starts[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
# include the upload's pyramid kernel in front
for r in rows[i0 - 2:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60]))
print("frame period %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
PY
