#!/bin/bash
# cfg.coarse A/B (gn_fca_coarse for the small levels vs one launch per iteration), interleaved. usage: tools/dbg/coarse_ab.sh OUTDIR
OUT=gpurun_out/$1; mkdir -p $OUT
run() {
  local name=$1; shift
  python3 bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --blocks 5 "$@" > $OUT/$name.json 2>> $OUT/err.log
  python3 - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s value %.3fM ms/step %.4f median-of-blocks %.4f" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], d.get("repeat_blocks", {}).get("ms_per_step_median", 0)), flush=True)
PY
}
for rep in 1 2; do
  run coarse_on_$rep --coarse 1
  run coarse_off_$rep --coarse 0
done
run on_c1_i3 --coarse 1 --coalesce 1 --inflight 3
run off_c1_i3 --coarse 0 --coalesce 1 --inflight 3
run on_c1_i1 --coarse 1 --coalesce 1 --inflight 1
run off_c1_i1 --coarse 0 --coalesce 1 --inflight 1
