#!/usr/bin/env python3
"""PCIe-inclusive rate of the batch workload: every step uploads all B keyframes (u8 image + f32 depth and
variance) and all B current frames from host memory before it aligns them — the case of a caller that keeps nothing
resident. Prints one JSON line; bench.py's `value` never includes these transfers."""
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

W, H, L, B = 640, 480, 4, 32
sched = [4, 7, 9, 12]
fx, fy, cx, cy = synth.default_intrinsics(W, H)
pairs = [synth.make_pair(W, H, seed=0x5EED + i) for i in range(4)]
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_iter=sched, max_keyframes=B,
                                     max_frames=B, max_batch=B))
slots = np.arange(B, dtype=np.int32)


def step(upload_keyframes, upload_frames):
    for b in range(B):
        p = pairs[b % len(pairs)]
        if upload_keyframes:
            ctx.keyframe_upload(b, p["kf_image"])
            ctx.keyframe_set_depth(b, p["depth0"], p["var0"])
        if upload_frames:
            ctx.frame_upload(b, p["cur_image"])
    return ctx.align(slots, slots)


def timed(kf, fr, n=10):
    step(True, True)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        step(kf, fr)
    ctx.sync()
    return (time.perf_counter() - t0) / n


out = {}
for name, kf, fr in (("resident", False, False), ("frames_uploaded_per_step", False, True), ("everything_uploaded_per_step", True, True)):
    d = timed(kf, fr)
    out[name] = {"ms_per_step": 1e3 * d, "gn_iterations_per_s": B * sum(sched) / d, "alignments_per_s": B / d}
out["bytes_uploaded_per_step"] = {"frames": B * W * H, "keyframes": B * (W * H + 2 * 4 * W * H)}
print(json.dumps(out))
ctx.close()
