#!/bin/bash
# Batch-stream count A/B: the shipping build (3 streams) and build/libellc_hip_s6.so (6), with and without GPU_MAX_HW_QUEUES=8
# (the HIP runtime's cap on hardware queues per process, default 4). usage (GPU box, repo root): tools/dbg/streams_ab.sh OUTDIR
OUT=gpurun_out/$1; mkdir -p $OUT
run() {  # name, queues, args
  local name=$1; shift; local q=$1; shift
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  python3 bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --blocks 5 --sustained 1000 "$@" > $OUT/$name.json 2>> $OUT/err.log
  python3 - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s value %.3fM ms/step %.4f sustained %.3fM" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d.get("sustained", {}).get("value", 0) / 1e6), flush=True)
PY
}
S6="--lib $PWD/build/libellc_hip_s6.so --streams 6"
run tree_q4 ""
run tree_q8 8
run s6_q4_c4_i16 "" $S6
run s6_q8_c4_i16 8 $S6
run s6_q8_c4_i24 8 $S6 --inflight 24
run s6_q8_c2_i12 8 $S6 --coalesce 2 --inflight 12
run s6_q8_c2_i14 8 $S6 --coalesce 2 --inflight 14
run s6_q8_c3_i18 8 $S6 --coalesce 3 --inflight 18
run s6_q16_c2_i14 16 $S6 --coalesce 2 --inflight 14
