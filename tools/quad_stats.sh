#!/bin/bash
# r06: share of gn_fca_dense4's quads whose four tap neighbourhoods do not fit one window (they take the per-pixel queue), on the C4
# level-0 launch at the converged pose, and per level over whole alignments. Needs the statistics variant:
#   make -C egomotion_with_local_loop_closures_amd/csrc variant VARDIR=variants NAME=qstats DEFS=-DELLC_QUAD_STATS
# usage (GPU box, repo root): tools/quad_stats.sh OUTNAME
OUT=gpurun_out/${1:-quad_stats}; mkdir -p $OUT
export ELLC_LIB_PATH=$PWD/egomotion_with_local_loop_closures_amd/csrc/variants/libellc_hip_qstats.so
for lv in 0 1 2 3 4; do
  python3 tools/profile_kernel.py --arith fast --width 1280 --height 960 --levels 5 --dense --batch 16 --reps 10 --level $lv > $OUT/level$lv.json 2> $OUT/level$lv.err
  echo "level $lv: $(grep ELLC_QUAD_STATS $OUT/level$lv.err)"
done
python3 - $OUT <<'PY'
import re, sys, json, os
out = {}
for lv in range(5):
    t = open(os.path.join(sys.argv[1], "level%d.err" % lv)).read()
    m = re.search(r"quads (\d+) nofit (\d+) .* queued_pixels (\d+) drain_rounds (\d+)", t)
    if m:
        q, nf, px, dr = (int(x) for x in m.groups())
        out["level%d" % lv] = {"quads": q, "quads_that_did_not_fit": nf, "share_of_quads": nf / q, "pixels_queued": px, "share_of_pixels": px / (4.0 * q), "drain_rounds": dr}
json.dump({"workload": "C4 (1280x960 dense x 64 alignments per launch, two scenes), gn_fca_dense4 at the converged pose, every launch of tools/profile_kernel.py",
           "levels": out}, open(os.path.join(sys.argv[1], "quad_stats.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
