export ELLC_LIB_PATH=$PWD/build/libellc_hip_diag.so
for N in "" "8" "8,8" "8,4,2" "6" "16"; do
  for rep in 1 2; do
  echo -n "bench NBLK='$N': "; ELLC_NBLK=$N python3 bench.py --lib $ELLC_LIB_PATH --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.3fM ms/step %.4f k0 %.1f us' % (d['value']/1e6, d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))"
  done
done
for N in "" "16" "16,16" "12"; do
  echo -n "c4 NBLK='$N': "; ELLC_NBLK=$N python3 bench.py --lib $ELLC_LIB_PATH --steps 12 --warmup 3 --no-extras --no-cpu-baseline --width 1280 --height 960 --levels 5 --dense --batch 16 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.3fM ms/step %.4f' % (d['value']/1e6, d['ms_per_step']))"
done
