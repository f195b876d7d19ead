#!/usr/bin/env python3
"""Host time of one ellc_frame_upload call (staging copy + the enqueues), with the device idle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import diaglib  # noqa: E402,F401
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
W, H, L = 640, 480, 4
pair = synth.make_pair(W, H, seed=31)
fx, fy, cx, cy = pair["intrinsics"]
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=2, max_frames=2, device=0))
img = pair["cur_image"]
for f in range(50): ctx.frame_upload(f & 1, img)
ctx.sync()
for label, gap in (("back to back", 0.0), ("200 us apart", 200e-6)):
    tot = 0.0
    for f in range(300):
        if gap:
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < gap: pass
        t0 = time.perf_counter(); ctx.frame_upload(f & 1, img); tot += time.perf_counter() - t0
    ctx.sync()
    print("frame_upload host time, %s: %.1f us" % (label, 1e6 * tot / 300))
buf = np.empty_like(img)
t0 = time.perf_counter()
for _ in range(300): np.copyto(buf, img)
print("memcpy of the image alone: %.1f us" % (1e6 * (time.perf_counter() - t0) / 300))
