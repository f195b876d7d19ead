OUT=gpurun_out/r03/smallmotion; mkdir -p $OUT
export ELLC_LIB_PATH=$PWD/build/libellc_hip_diag.so
for M in "0.002 0.004" "0.005 0.01" "0.01 0.02"; do set -- $M
 for NW in 0 1; do
  if [ $NW = 1 ]; then export ELLC_NO_WINDOWS=1; else unset ELLC_NO_WINDOWS; fi
  python3 tools/profile_kernel.py --arith fast --rot $1 --trans $2 > $OUT/k640_r$1_nw$NW.json 2>>$OUT/err.log
  python3 tools/profile_kernel.py --arith fast --rot $1 --trans $2 --width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10 > $OUT/kc4_r$1_nw$NW.json 2>>$OUT/err.log
 done
done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "k*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-28s avg_us %7.1f  frac %.3f" % (os.path.basename(f), 1e3 * d["avg_ms"], d["achieved_GBps"] / 8000))
PY
