#!/bin/bash
# The fused kernel at every level of the bench workload (640x480 x 128 alignments, HIP events) under each library given.
# usage (GPU box, repo root): tools/ab_levels.sh OUTDIR ROUNDS lib1.so lib2.so ...   ("tree" = the shipping library)
OUT=gpurun_out/$1; ROUNDS=$2; shift 2
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" = tree ]; then unset ELLC_LIB_PATH; name=tree; else export ELLC_LIB_PATH=$PWD/$lib; name=$(basename $lib .so); fi
    for lv in 0 1 2 3; do
      python3 tools/profile_kernel.py --arith ${ARITH:-fast} --level $lv > $OUT/k640l${lv}_${name}_$r.json 2>> $OUT/err.log
    done
  done
done
python3 - $OUT <<'PY'
import json, sys, glob, os, statistics
by = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "k*_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    key = os.path.basename(f).rsplit("_", 1)[0]
    by.setdefault(key, []).append(1e3 * d["avg_ms"])
for k, v in sorted(by.items()):
    print("%-40s median %7.1f us  (%s)" % (k, statistics.median(v), " ".join("%.1f" % x for x in v)))
PY
