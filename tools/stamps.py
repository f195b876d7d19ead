#!/usr/bin/env python3
"""Diagnostic: where one fused launch spends its time (block (0,0), 100 MHz s_memrealtime stamps). Needs the
STAMPS build: ELLC_LIB_PATH=build/libellc_hip_stamps.so python tools/stamps.py --level 3"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth, _lib  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)

ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=3)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--arith", choices=["fast", "exact"], default="fast")
a = ap.parse_args()
W, H, L, B = 640, 480, 4, a.batch
fx, fy, cx, cy = synth.default_intrinsics(W, H)
pairs = [synth.make_pair(W, H, seed=0x5EED + i) for i in range(4)]
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_keyframes=B, max_frames=B, max_batch=B,
                                     arith=api.ARITH_FAST if a.arith == "fast" else api.ARITH_EXACT))
for b in range(B):
    p = pairs[b % 4]
    ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"]); ctx.frame_upload(b, p["cur_image"])
slots = np.arange(B, dtype=np.int32)
names = ["start", "partials loaded", "combined", "LU done", "delta done", "se3 done", "prologue end", "pixels done", "reduced+stored"]
acc = np.zeros(9)
n = 0
for r in range(10):
    ms, _, _ = ctx.profile_gn_kernel(slots, slots, a.level, reps=3)
    st = (C.c_ulonglong * 64)()
    _lib.lib().ellc_debug_stamps(ctx.h, st)
    t = np.array([st[i] for i in range(9)], dtype=np.float64)
    d = (t - t[0]) * 0.01   # 100 MHz -> us
    acc += d
    n += 1
print("level %d, B=%d: launch avg %.2f us (HIP events)" % (a.level, B, ms * 1e3))
print("  block (0,0), us since its start: " + ", ".join("%s %.2f" % (names[i], acc[i] / n) for i in range(1, 9)))
nb = 1024 if a.level == 0 else 256
bs = (C.c_ulonglong * (4 * nb))()
_lib.lib().ellc_debug_block_stamps(ctx.h, bs, nb)
t = np.array(list(bs), dtype=np.float64).reshape(nb, 4) * 0.01
t0 = t[:, 0].min()
print("  per-block (us since first block start): start min/med/max %.2f %.2f %.2f | prologue end %.2f %.2f %.2f | pixels done %.2f %.2f %.2f | end %.2f %.2f %.2f" % (
    (t[:, 0] - t0).min(), np.median(t[:, 0] - t0), (t[:, 0] - t0).max(), (t[:, 1] - t0).min(), np.median(t[:, 1] - t0), (t[:, 1] - t0).max(),
    (t[:, 2] - t0).min(), np.median(t[:, 2] - t0), (t[:, 2] - t0).max(), (t[:, 3] - t0).min(), np.median(t[:, 3] - t0), (t[:, 3] - t0).max()))
dur = t[:, 2] - t[:, 1]
print("  pixel phase per block: min %.2f med %.2f max %.2f" % (dur.min(), np.median(dur), dur.max()))
bid = np.arange(nb)
print("  by blockIdx.x (tile phase) :", np.round([dur[(bid % 32) == k].mean() for k in range(0, 32, 4)], 1))
print("  by blockIdx.y (alignment)  :", np.round([dur[(bid // 32) == k].mean() for k in range(0, 32, 4)], 1))
print("  by bid // 256 (dispatch round): pixel phase", np.round([dur[(bid // 256) == k].mean() for k in range(nb // 256)], 1),
      "| pixels-done time", np.round([(t[(bid // 256) == k, 2] - t0).mean() for k in range(nb // 256)], 1))
print("  by bid %% 8 (XCD group)     :", np.round([dur[(bid % 8) == k].mean() for k in range(8)], 1))
order = np.argsort(dur)
print("  slowest 12 blocks:", order[-12:], np.round(dur[order[-12:]], 1))
print("  fastest 12 blocks:", order[:12], np.round(dur[order[:12]], 1))
print("  start-time of slowest/fastest:", np.round((t[order[-12:], 0] - t0), 2), np.round((t[order[:12], 0] - t0), 2))
prev = 0.0
for nm, v in zip(names, acc / n):
    print("  %-18s t=%6.2f us  (+%5.2f)" % (nm, v, v - prev))
    prev = v
