#!/bin/bash
# Level-0 kernel timings (HIP events) under each library build given: tree (the shipping library) or build/libellc_hip_<name>.so.
# usage (GPU box, repo root): tools/ab_kernels.sh OUTDIR name1 name2 ...
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = tree ]; then unset ELLC_LIB_PATH; LIBARG=""; else export ELLC_LIB_PATH=$PWD/build/libellc_hip_$v.so; LIBARG="--lib $ELLC_LIB_PATH"; fi
  for rep in 1 2; do
    python3 tools/profile_kernel.py --arith fast > $OUT/k640_${v}_$rep.json 2>> $OUT/err.log
    python3 tools/profile_kernel.py --arith fast --width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10 > $OUT/kc4_${v}_$rep.json 2>> $OUT/err.log
  done
  python3 bench.py $LIBARG --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_$v.json 2>> $OUT/err.log
done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "k*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-28s avg_us %7.1f  frac %.3f" % (os.path.basename(f), 1e3 * d["avg_ms"], d["achieved_GBps"] / 8000))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-28s value %.3fM  ms/step %.4f  k0 %.1f us" % (os.path.basename(f), d["value"] / 1e6, d["ms_per_step"], 1e3 * d["roofline"]["avg_launch_ms"]))
PY
