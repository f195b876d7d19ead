#!/usr/bin/env python3
"""Interleaved A/B of accumulate-kernel variants inside ONE process (cdna guide rule 24): one context per variant
(the variant is read from the environment when the context is created), alternating timed rounds."""
import argparse
import os
import sys
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth, _lib  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)
import ctypes  # noqa: E402

# an older library build (ELLC_LIB_PATH) may lack entry points added since; the timing hooks used here are old
_so = ctypes.CDLL(_lib.SO_PATH)
_lib.ABI_SYMBOLS = [s for s in _lib.ABI_SYMBOLS if hasattr(_so, s)]

ap = argparse.ArgumentParser()
ap.add_argument("--variants", nargs="+", default=["X=1"], help="environment settings, one context each: \"A=1;B=2\"")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--level", type=int, default=0)
ap.add_argument("--dense", action="store_true")
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
a = ap.parse_args()
W, H, L, B = a.width, a.height, 4, a.batch
fx, fy, cx, cy = synth.default_intrinsics(W, H)
pairs = [synth.make_pair(W, H, seed=0x5EED + i, dense=a.dense) for i in range(min(8, B))]
ctxs = []
for v in a.variants:
    keys = []
    for kv in v.split(";"):
        if kv:
            k, val = kv.split("=")
            os.environ[k] = val
            keys.append(k)
    ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_keyframes=B, max_frames=B, max_batch=B))
    for k in keys:
        del os.environ[k]
    for b in range(B):
        p = pairs[b % len(pairs)]
        ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"]); ctx.frame_upload(b, p["cur_image"])
    ctxs.append(ctx)
slots = np.arange(B, dtype=np.int32)
res = {v: [] for v in a.variants}
full = {v: [] for v in a.variants}
for r in range(a.rounds):
    for v, ctx in zip(a.variants, ctxs):
        ms, alg, V = ctx.profile_gn_kernel(slots, slots, a.level, reps=20)
        res[v].append(ms)
        full[v].append(ctx.profile_align(slots, slots, reps=5))
for v in a.variants:
    print("%-40s kernel L%d: median %.4f ms min %.4f | full align: median %.4f ms min %.4f" % (
        v, a.level, statistics.median(res[v]), min(res[v]), statistics.median(full[v]), min(full[v])))
poses = [ctx.align(slots, slots)[0] for ctx in ctxs]
print("max |pose difference| between variants: %.3e" % max(float(np.abs(p - poses[0]).max()) for p in poses))
