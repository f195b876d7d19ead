#!/usr/bin/env python3
"""Drives one keyframe's depth-map cycle (observe, fill holes, regularise, export; then createKeyFrame) repeatedly at
640x480 so rocprofv3 --kernel-trace --stats can time the depth kernels; also prints host wall times per stage."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)

W, H, L = 640, 480, 4
pair = synth.make_pair(W, H, seed=31, rot=0.006, trans=0.03)
fx, fy, cx, cy = pair["intrinsics"]
st = synth.make_depth_state(W, H, 9, pair["kf_image"], pair["idepth_true"])
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, max_keyframes=2, max_frames=1))
ctx.keyframe_upload(0, pair["kf_image"]); ctx.frame_upload(0, pair["cur_image"]); ctx.keyframe_from_frame(1, 0)
xi = pair["xi_true"]
reps = 20
t = {}


def timed(name, fn):
    ctx.sync()
    t0 = time.perf_counter()
    fn()
    ctx.sync()
    t[name] = t.get(name, 0.0) + time.perf_counter() - t0


for r in range(reps):
    ctx.depth_set_keyframe(0); ctx.depth_set_state(st)
    timed("regularize", lambda: ctx.depth_regularize(False))
    timed("observe (line stereo)", lambda: ctx.depth_observe(0, xi))
    timed("fill_holes", lambda: ctx.depth_fill_holes())
    timed("regularize", lambda: ctx.depth_regularize(False))
    timed("update_depth_image (+pyramid)", lambda: ctx.depth_update_depth_image())
    timed("do_regularization (fill + regularise, one launch)", lambda: ctx.depth_do_regularization(False))
    timed("create_keyframe (propagate..export)", lambda: ctx.depth_create_keyframe(1, xi))
print("valid hypotheses: %d of %d" % (int(st["valid"].sum()), W * H))
for k, v in t.items():
    n = reps * (2 if k == "regularize" else 1)
    print("%-52s %8.1f us per call (host wall incl. launch + sync)" % (k, 1e6 * v / n))
ctx.close()
