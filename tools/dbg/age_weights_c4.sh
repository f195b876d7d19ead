#!/bin/bash
export ELLC_LIB_PATH=$PWD/egomotion_with_local_loop_closures_amd/csrc/variants/libellc_hip_envdiag.so
C4="--arith fast --width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10"
for r in 1 2; do
for w in "3:1.2,1.0,0.8" "3:1,1,1" "3:1.1,1.0,0.9" "3:1.3,1.0,0.7" "3:1.15,1.0,0.85" "3:1.25,1.05,0.7"; do
  ELLC_AGE_W="$w" python3 tools/profile_kernel.py $C4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(1e3*d['avg_ms'],1))"
done
ELLC_NO_AGE_BALANCE=1 python3 tools/profile_kernel.py $C4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no age balance', round(1e3*d['avg_ms'],1))"
done
