#!/bin/bash
# Kernel trace + stats of the timed workload of bench.py (GPU box, repo root): per-kernel totals and per-(kernel, grid) medians.
# usage: tools/dbg/trace_bench.sh OUTNAME [bench flags]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --trace-only "$@" > $OUT/trace.log 2>&1
echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/trace_summary.py $OUT/trace
f=$(ls $OUT/trace/*/*_kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %6s total %10.1f us avg %8.2f us  %5.1f %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
python3 tools/dbg/chain_gaps.py $OUT/trace; rm -rf $OUT/trace
