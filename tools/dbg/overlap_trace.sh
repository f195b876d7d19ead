#!/bin/bash
# Kernel trace of the pipelined bench + overlap summary (busy share, kernels in flight, time per family).
# usage: tools/dbg/overlap_trace.sh OUTDIR [bench args]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 240 --warmup 8 --trace-only --no-cpu-baseline "$@" > $OUT/run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/overlap_summary.py $OUT 128 | tee $OUT/overlap.txt
