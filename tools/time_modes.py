#!/usr/bin/env python3
"""Device time of one full ellc_align (HIP events, graph replay) for both schedules — FCA (tracking) and ICA (loop-closure
batch) — at B = 32 and B = 1, 640x480 semi-dense."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)
W, H, L, B = 640, 480, 4, 32
fx, fy, cx, cy = synth.default_intrinsics(W, H)
pairs = [synth.make_pair(W, H, seed=0x5EED + i) for i in range(4)]
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_keyframes=B, max_frames=B, max_batch=B))
for b in range(B):
    p = pairs[b % 4]
    ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"]); ctx.frame_upload(b, p["cur_image"])
slots = np.arange(B, dtype=np.int32)
# saved weights for ICA: run FCA with save_weights once, finalise
ctx.align(slots, slots, save_weights=True)
for b in range(B):
    ctx.keyframe_finalise_weights(b)
for mode in (api.MODE_FCA, api.MODE_ICA):
    for Bn in (32, 1):
        ms = [ctx.profile_align(slots[:Bn], slots[:Bn], mode=mode, reps=5) for _ in range(5)]
        print("mode", mode, "B", Bn, "median ms %.4f" % np.median(ms))
