#!/usr/bin/env python3
"""Where an iteration inside gn_fca_coarse goes (block 0, 100 MHz stamps; build: make -C csrc stamps).
usage: coarse_stamps.py [B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth, _lib  # noqa: E402
_lib.use_library(os.path.join(ROOT, "build", os.environ.get("ELLC_STAMPS_LIB", "libellc_hip_stamps.so")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H, L = 640, 480, 4
pairs = synth.make_loop_closure_batch(W, H, min(B, 8), seed=5)
fx, fy, cx, cy = pairs[0]["intrinsics"]
for arith in (api.ARITH_FAST, api.ARITH_EXACT):
    ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, max_keyframes=B, max_frames=1, max_batch=B, arith=arith))
    ctx.frame_upload(0, pairs[0]["cur_image"])
    for b in range(B):
        p = pairs[b % len(pairs)]
        ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"])
    kf = np.arange(B); fr = np.zeros(B, np.int64)
    acc = np.zeros((15, 4)); n = 0
    for r in range(20):
        ctx.align(kf, fr)
        st = (C.c_ulonglong * 64)()
        _lib.lib().ellc_debug_stamps(ctx.h, st)
        t = np.array([st[i] for i in range(60)], dtype=np.float64).reshape(15, 4) * 0.01
        if r >= 5:
            acc += t - t[0, 0]; n += 1
    acc /= n
    print("arith", "fast" if arith == api.ARITH_FAST else "exact", "B", B)
    for i in range(15):
        nxt = acc[i + 1, 0] - acc[i, 3] if i < 14 else 0.0
        print("  it %2d start %7.2f us | pixels %5.2f | reduce+combine %5.2f | solve %5.2f | to next %5.2f" % (
            i, acc[i, 0], acc[i, 1] - acc[i, 0], acc[i, 2] - acc[i, 1], acc[i, 3] - acc[i, 2], nxt))
    ctx.close()
