export ELLC_LIB_PATH=$PWD/build/libellc_hip_diag.so
for NW in 1 0; do
if [ $NW = 1 ]; then export ELLC_NO_WINDOWS=1; else unset ELLC_NO_WINDOWS; fi
for N in 8 16 32; do
  echo -n "c4 nowin=$NW nblk $N: "; ELLC_NBLK=$N python3 tools/profile_kernel.py --arith fast --width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f us frac %.3f' % (1e3*d['avg_ms'], d['achieved_GBps']/8000))"
done
for N in 4 8 16; do
  echo -n "640 nowin=$NW nblk $N: "; ELLC_NBLK=$N python3 tools/profile_kernel.py --arith fast 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f us frac %.3f' % (1e3*d['avg_ms'], d['achieved_GBps']/8000))"
done
done
