#!/bin/bash
# Per-kernel averages of the timed bench workload (rocprofv3 --kernel-trace --stats of bench.py --trace-only) under each library given, on
# one box: where a difference between two builds sits. usage (GPU box, repo root): tools/ab_trace.sh OUTNAME lib1.so lib2.so ...
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --lib $ROOT/$lib --steps 40 --warmup 40 --trace-only > $OUT/$name.log 2>&1
  echo "$name rc=$?"
done
python3 - $OUT "$@" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
tabs = {}
for lib in sys.argv[2:]:
    name = os.path.basename(lib)[:-3]
    f = glob.glob(os.path.join(out, name, "*", "*_kernel_stats.csv"))
    if not f:
        continue
    tabs[name] = {r["Name"].split("(")[0].replace("void ", "").replace("ellc::", ""): (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f[0]))}
keys = sorted(set(k for t in tabs.values() for k in t))
print("%-44s" % "kernel" + "".join("%26s" % n[-24:] for n in tabs))
for k in keys:
    print("%-44s" % k[:44] + "".join("%26s" % ("%d x %.1f us" % tabs[n][k] if k in tabs[n] else "-") for n in tabs))
PY
