#!/bin/bash
# One early-exit alignment (fast mode, resident schedule) and the tracked frame under block counts per level given through the env knob
# of the -DELLC_DIAG build (make -C .../csrc diag): usage (GPU box, repo root): tools/dbg/sweep_persist_nblk.sh OUT "256,120,30,7" "128,120,30,7" ...
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export ELLC_LIB_PATH=$PWD/build/libellc_hip_envdiag.so
for rep in 1 2; do
  for nb in "$@"; do
    export ELLC_NBLK=$nb
    echo "rep $rep ELLC_NBLK=$nb: $(python3 tools/dbg/single_ee.py 2>>$OUT/err.log | grep 'fast save_weights 0 schedule mode 1' | sed 's/.*mode 1: //')  | track: $(python3 tools/dbg/track_ab.py 3 2>>$OUT/err.log)" | tee -a $OUT/sweep.txt
  done
done
