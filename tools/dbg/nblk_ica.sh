#!/bin/bash
export ELLC_LIB_PATH=$PWD/egomotion_with_local_loop_closures_amd/csrc/variants/libellc_hip_envdiag.so
for rep in 1 2; do
  for p in 8,4,4,2 16,4,4,2 16,8,4,2 24,8,4,2 32,8,4,2 32,16,4,2 48,16,8,2; do
    b=$(ELLC_NBLK=$p python3 bench.py --mode ica --lib $ELLC_LIB_PATH --no-extras --no-cpu-baseline --blocks 0 --sustained 0 --steps 40 --warmup 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
    echo "ICA nblk $p rep $rep: $b ms per step"
  done
done
