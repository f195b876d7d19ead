#!/usr/bin/env python3
"""BASELINE configs[1] as the reference's main loop runs it (main.cpp:330, 499-502): per frame — upload + pyramid, one
alignment against the active keyframe (FCA, early exit on), then the depth refinement of the keyframe with the new pose
(observe, fill holes, regularise, export to the keyframe's depth pyramid). Only the alignment's result is waited for."""
import json
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
ARITH = sys.argv[2] if len(sys.argv) > 2 else "fast"   # usage: bench_track.py [frames] [fast|exact] [fused]
FUSED = len(sys.argv) > 3 and sys.argv[3] == "fused"
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=2, max_frames=2,
                                     arith=api.ARITH_FAST if ARITH == "fast" else api.ARITH_EXACT))
ctx.keyframe_upload(0, pair["kf_image"])
ctx.depth_set_keyframe(0)
ctx.depth_set_state(st)
ctx.depth_update_depth_image()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t = {"upload": 0.0, "align": 0.0, "depth": 0.0}
iters = 0
ctx.sync()
t00 = time.perf_counter()
for f in range(N):
    t0 = time.perf_counter()
    ctx.frame_upload(f & 1, pair["cur_image"])
    t1 = time.perf_counter()
    if FUSED:   # one ellc_track_frame call per frame: the depth stages behind the alignment on the device, saved weights
        pose, it, _, _ = ctx.track_frame(f & 1, save_weights=True)
        t2 = time.perf_counter()
    else:
        pose, it, _ = ctx.align([0], [f & 1])
        t2 = time.perf_counter()
        ctx.depth_observe(f & 1, pose[0])
        ctx.depth_fill_holes()
        ctx.depth_regularize(False)
        ctx.depth_update_depth_image()
    t3 = time.perf_counter()
    t["upload"] += t1 - t0; t["align"] += t2 - t1; t["depth"] += t3 - t2
    iters += int(np.asarray(it).sum())
ctx.sync()
dt = time.perf_counter() - t00
print(json.dumps({"arith": ARITH, "frames": N, "ms_per_frame": 1e3 * dt / N, "frames_per_s": N / dt, "mean_gn_iterations": iters / N,
                  "host_ms": {k: 1e3 * v / N for k, v in t.items()}}))
ctx.close()
