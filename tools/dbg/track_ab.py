#!/usr/bin/env python3
"""bench.py's tracked-frame record for the library ELLC_LIB_PATH names (A/B of builds on one box; box-to-box the figure moves by
2 %): best of n runs, ms per frame, five calls per frame / one ellc_track_frame call per frame. usage: track_ab.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import diaglib  # noqa: E402,F401
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402


class A:
    arith = "fast"


best = [9.0, 9.0]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    r = bench.tracked_frame(api, synth, A, 0)
    best = [min(best[0], r["ms_per_frame"]), min(best[1], r["ms_per_frame_fused_call"])]
name = os.path.basename(os.environ.get("ELLC_LIB_PATH", "tree")).replace("libellc_hip_", "").replace(".so", "")
print("%-10s five calls %.4f ms   one call %.4f ms" % (name, best[0], best[1]))
