#!/usr/bin/env python3
"""Per-iteration cost of the coarse levels: ms per ellc_align call for schedules that differ in the coarse levels' iteration
counts, with gn_fca_coarse (cfg.coarse = 0) and with one launch per iteration (-1). usage: coarse_iter.py [fast|exact]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
arith = api.ARITH_FAST if (len(sys.argv) < 2 or sys.argv[1] == "fast") else api.ARITH_EXACT
W, H, L = 640, 480, 4
for B in (1, 32, 64):
    pairs = synth.make_loop_closure_batch(W, H, min(B, 8), seed=5)
    fx, fy, cx, cy = pairs[0]["intrinsics"]
    for sched in ([4, 7, 9, 12], [0, 0, 0, 12], [0, 0, 0, 2], [0, 0, 9, 0], [0, 0, 2, 0]):
        row = []
        for coarse in (1, -1, 0):
            cfg = api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, max_iter=sched, max_keyframes=B, max_frames=1, max_batch=B, arith=arith,
                                     coarse=coarse)
            ctx = api.Context(cfg)
            ctx.frame_upload(0, pairs[0]["cur_image"])
            for b in range(B):
                p = pairs[b % len(pairs)]
                ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"])
            kf = np.arange(B); fr = np.zeros(B, np.int64)
            for _ in range(10):
                ctx.align(kf, fr)
            ctx.sync()
            t0 = time.perf_counter()
            n = 100
            for _ in range(n):
                ctx.align(kf, fr)
            row.append((time.perf_counter() - t0) / n * 1e3)
            ctx.close()
        print("B %3d sched %-16s kernel %.4f ms | launches, kernel's block counts %.4f ms | launches, default block counts %.4f ms" % (B, sched, row[0], row[1], row[2]), flush=True)
