#!/bin/bash
# C4 level-0 kernel only (HIP events, tools/profile_kernel.py, 1280x960 dense x 64) under each library given, interleaved over ROUNDS
# rounds on one box. usage (GPU box, repo root): tools/ab_c4.sh OUTDIR ROUNDS lib1.so lib2.so ...   ("tree" = the in-tree diagnostic library)
OUT=gpurun_out/$1; ROUNDS=$2; shift 2
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" = tree ]; then unset ELLC_LIB_PATH; name=tree; else export ELLC_LIB_PATH=$PWD/$lib; name=$(basename $lib .so); fi
    python3 tools/profile_kernel.py --arith ${ARITH:-fast} --width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10 > $OUT/kc4_${name}_$r.json 2>> $OUT/err.log
  done
done
python3 - $OUT <<'PY'
import json, sys, glob, os, statistics
by = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "kc4_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    key = os.path.basename(f).rsplit("_", 1)[0]
    by.setdefault(key, []).append(1e3 * d["avg_ms"])
for k, v in sorted(by.items()):
    print("%-40s median %7.1f us  (%s)" % (k, statistics.median(v), " ".join("%.1f" % x for x in v)))
PY
