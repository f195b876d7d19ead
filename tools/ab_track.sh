#!/bin/bash
# Tracked frame (tools/dbg/track_ab.py) and one early-exit alignment (tools/dbg/single_ee.py, fast mode rows) under each library given,
# interleaved over ROUNDS rounds on one box. usage (GPU box, repo root): tools/ab_track.sh OUTDIR ROUNDS lib1.so lib2.so ...  ("tree" = in-tree)
OUT=gpurun_out/$1; ROUNDS=$2; shift 2
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" = tree ]; then unset ELLC_LIB_PATH; name=tree; else export ELLC_LIB_PATH=$PWD/$lib; name=$(basename $lib .so); fi
    echo "round $r $name: $(python3 tools/dbg/track_ab.py 3 2>>$OUT/err.log)" | tee -a $OUT/track.txt
    python3 tools/dbg/single_ee.py 2>>$OUT/err.log | sed "s/^/round $r $name: /" | tee -a $OUT/single.txt
  done
done
