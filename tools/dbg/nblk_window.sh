#!/bin/bash
export ELLC_LIB_PATH=$PWD/egomotion_with_local_loop_closures_amd/csrc/variants/libellc_hip_envdiag.so
for rep in 1 2 3; do
  for p in 8,4,4,2 8,8,4,2 8,8,8,4 8,6,4,2 8,8,4,4; do
    b=$(ELLC_NBLK=$p python3 bench.py --lib $ELLC_LIB_PATH --no-extras --no-cpu-baseline --blocks 0 --sustained 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
    echo "nblk $p rep $rep: driver window $b ms per step"
  done
done
