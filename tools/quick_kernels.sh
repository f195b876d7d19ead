#!/bin/bash
# Level-0 kernel timings (HIP events, tools/profile_kernel.py) of the bench workload and of C4, both arithmetic modes, then the
# contract fields of bench.py. usage (GPU box, repo root): tools/quick_kernels.sh OUTDIR
OUT=gpurun_out/${1:-quick}
mkdir -p $OUT
for A in fast exact; do
  python3 tools/profile_kernel.py --arith $A > $OUT/k640_$A.json 2>> $OUT/err.log
  python3 tools/profile_kernel.py --arith $A --width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10 > $OUT/kc4_$A.json 2>> $OUT/err.log
done
python3 tools/profile_kernel.py --arith fast --level 1 > $OUT/k640_fast_l1.json 2>> $OUT/err.log
python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_fast.json 2>> $OUT/err.log
python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --arith exact > $OUT/bench_exact.json 2>> $OUT/err.log
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "k*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "avg_us %.1f" % (1e3 * d["avg_ms"]), "GB/s %.0f" % d["achieved_GBps"], "frac %.3f" % (d["achieved_GBps"] / 8000))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "value %.3fM" % (d["value"] / 1e6), "ms/step %.4f" % d["ms_per_step"], "k0 us %.1f frac %.3f" % (1e3 * d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
