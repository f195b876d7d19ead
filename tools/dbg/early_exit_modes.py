#!/usr/bin/env python3
"""Early-exit FCA batches: state-driven schedule (production) against the level-bound one (ELLC_NO_ADAPTIVE=1, diagnostic
build: ELLC_LIB_PATH=build/libellc_hip_envdiag.so; ELLC_ADAPTIVE_MAX_BATCH=32 lifts the batch-size limit), per batch size. Prints ms per batch and the iteration totals per alignment."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)

W, H, L = 640, 480, 4
fx, fy, cx, cy = synth.default_intrinsics(W, H)
for B in (1, 2, 4, 8, 32):
    views = synth.make_shared_frame_batch(W, H, B, seed=0xE11C)
    variants = (("adaptive", {}), ("adaptive, first graph 20", {"ELLC_ADAPTIVE_FIRST": "20"}), ("level-bound", {"ELLC_NO_ADAPTIVE": "1"}))
    ctxs = {}
    for name, env in variants:
        for k in ("ELLC_NO_ADAPTIVE", "ELLC_ADAPTIVE_FIRST"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=B, max_frames=1, max_batch=B,
                                             arith=api.ARITH_FAST))
        ctx.frame_upload(0, views[0]["cur_image"])
        for i in range(B):
            ctx.keyframe_upload(i, views[i]["kf_image"])
            ctx.keyframe_set_depth(i, views[i]["depth0"], views[i]["var0"])
        ctxs[name] = ctx
    kf, fr = list(range(B)), [0] * B
    res = {name: ctx.align(kf, fr) for name, ctx in ctxs.items()}
    best = {name: 1e9 for name in ctxs}
    for _ in range(5):   # alternating rounds in one process, the fastest round of each
        for name, ctx in ctxs.items():
            for _ in range(3):
                ctx.align(kf, fr)
            t0 = time.perf_counter()
            for _ in range(40):
                ctx.align(kf, fr)
            best[name] = min(best[name], (time.perf_counter() - t0) / 40 * 1e3)
    tot = res["adaptive"][1].sum(axis=1)
    same = np.array_equal(res["adaptive"][0], res["level-bound"][0]) and np.array_equal(res["adaptive"][1], res["level-bound"][1])
    print(f"B={B}: " + ", ".join(f"{n} {v:.4f} ms" for n, v in best.items()) + f"; identical results {same}; "
          f"iterations per alignment min {tot.min()} mean {tot.mean():.1f} max {tot.max()}", flush=True)
    for ctx in ctxs.values():
        ctx.close()
