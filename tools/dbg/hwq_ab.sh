#!/bin/bash
# GPU_MAX_HW_QUEUES (the HIP runtime's cap on hardware queues per process, default 4) against the pipelined bench, interleaved.
# usage (GPU box, repo root): tools/dbg/hwq_ab.sh OUTDIR
OUT=gpurun_out/$1; mkdir -p $OUT
run() {
  local name=$1; shift; local q=$1; shift
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" > $OUT/$name.json 2>> $OUT/err.log
  python3 - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-12s value %.3fM ms/step %.4f" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"]), flush=True)
PY
}
for rep in 1 2 3 4; do
  run q4_$rep ""
  run q8_$rep 8
  run q6_$rep 6
done
