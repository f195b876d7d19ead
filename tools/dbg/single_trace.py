#!/usr/bin/env python3
"""One alignment at a time (C1: B = 1, 640x480, 4 levels, full schedule), N times: the process rocprofv3 wraps to see what a single
alignment's launches cost (ELLC_LIB_PATH names the build). usage: single_trace.py [n] [early_exit]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from egomotion_with_local_loop_closures_amd import api, synth
import diaglib  # noqa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ee = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H, L = 640, 480, 4
p = synth.make_pair(W, H, seed=7)
fx, fy, cx, cy = p["intrinsics"]
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=ee, max_keyframes=1, max_frames=1, max_batch=1, arith=api.ARITH_FAST))
ctx.keyframe_upload(0, p["kf_image"]); ctx.keyframe_set_depth(0, p["depth0"], p["var0"]); ctx.frame_upload(0, p["cur_image"])
for _ in range(100):
    ctx.align([0], [0])
ctx.sync()
t0 = time.perf_counter()
for _ in range(n):
    pose, it, w = ctx.align([0], [0])
ctx.sync()
print("ms per alignment %.4f  iters %s" % (1e3 * (time.perf_counter() - t0) / n, it[0]))
ctx.close()
