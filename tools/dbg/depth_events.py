#!/usr/bin/env python3
"""HIP-event times of the depth-map stages (bench.py's `depth` sub-record alone)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diaglib  # noqa: E402,F401
d = bench.depth_kernels(api, synth, 0)
for k, v in d["kernels"].items():
    print("%-40s %6.1f us  frac %.3f" % (k[:40], v["us_per_call"], v["frac_of_hbm_peak"]))
print("create_keyframe wall %.1f us" % d["create_keyframe"]["us_per_call_wall"])
