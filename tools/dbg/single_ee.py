#!/usr/bin/env python3
"""One 640x480 alignment with the reference's early exit (the tracking call's alignment: state-driven schedule), synchronous calls:
ms per call and the iterations it ran, with the schedule as one resident launch (mode 1, the default) and as one launch per
iteration (mode 0: ellc_ctx_set_persistent_schedule)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import diaglib  # noqa: F401
from egomotion_with_local_loop_closures_amd import api, synth
W, H, L = 640, 480, 4
fx, fy, cx, cy = synth.default_intrinsics(W, H)
for arith, sw, mode in [(a_, s_, m_) for a_ in ("fast", "exact") for s_ in (0, 1) for m_ in (0, 1)]:
    if True:
        p = synth.make_pair(W, H, seed=0x5EED)
        ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=1, max_frames=1, max_batch=1,
                                             arith=api.ARITH_FAST if arith == "fast" else api.ARITH_EXACT))
        ctx.set_persistent_schedule(mode)
        ctx.keyframe_upload(0, p["kf_image"]); ctx.keyframe_set_depth(0, p["depth0"], p["var0"]); ctx.frame_upload(0, p["cur_image"])
        kf = np.zeros(1, np.int32)
        for _ in range(50):
            pose, iters, _ = ctx.align(kf, kf, save_weights=bool(sw))
        ctx.sync(); t0 = time.perf_counter()
        n = 400
        for _ in range(n):
            pose, iters, _ = ctx.align(kf, kf, save_weights=bool(sw))
        ctx.sync(); dt = time.perf_counter() - t0
        print("%s save_weights %d schedule mode %d: %.4f ms per alignment, iterations %s" % (arith, sw, mode, 1e3 * dt / n, [int(v) for v in iters[0]]))
        ctx.close()
