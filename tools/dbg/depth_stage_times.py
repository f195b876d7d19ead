#!/usr/bin/env python3
"""Device time of each depth-map stage at 640x480 (ellc_profile_depth_stage, HIP events): the numbers of bench.py's `depth` record."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)

rec = bench.depth_kernels(api, synth, 0)
for k, v in rec["kernels"].items():
    print(f"{v['us_per_call']:8.2f} us  {k}")
print(f"{rec['create_keyframe']['us_per_call_wall']:8.2f} us  create_keyframe (wall)")
