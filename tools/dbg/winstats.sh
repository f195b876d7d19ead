export ELLC_LIB_PATH=$PWD/build/libellc_hip_diag.so
for M in "0.002 0.004" "0.01 0.02"; do set -- $M
  echo "== rot $1 trans $2 640"; python3 tools/profile_kernel.py --arith fast --rot $1 --trans $2 --reps 2 2>&1 | grep win_stats
  echo "== rot $1 trans $2 c4"; python3 tools/profile_kernel.py --arith fast --rot $1 --trans $2 --width 1280 --height 960 --levels 5 --dense --batch 16 --reps 2 2>&1 | grep win_stats
done
