#!/usr/bin/env python3
"""r06: many synchronous early-exit alignments (the tracking call's, one resident launch each) in a row — how many resident launches had
to be abandoned (ellc_debug_persist_counters), how many blocks re-joined through the state line, and the slowest calls.
usage: persist_soak.py [calls] [save_weights 0|1] [fast|exact]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import diaglib  # noqa: F401
from egomotion_with_local_loop_closures_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sw = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
arith = sys.argv[3] if len(sys.argv) > 3 else "fast"
W, H, L = 640, 480, 4
fx, fy, cx, cy = synth.default_intrinsics(W, H)
p = synth.make_pair(W, H, seed=0x5EED)
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=1, max_frames=1, max_batch=1,
                                     arith=api.ARITH_FAST if arith == "fast" else api.ARITH_EXACT))
ctx.keyframe_upload(0, p["kf_image"]); ctx.keyframe_set_depth(0, p["depth0"], p["var0"]); ctx.frame_upload(0, p["cur_image"])
kf = np.zeros(1, np.int32)
for _ in range(100):
    ref = ctx.align(kf, kf, save_weights=sw)
l0, a0, r0 = ctx.debug_persist_counters()
ts = np.zeros(n)
bad = 0
for i in range(n):
    t0 = time.perf_counter()
    out = ctx.align(kf, kf, save_weights=sw)
    ts[i] = time.perf_counter() - t0
    if not (np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])):
        bad += 1
l1, a1, r1 = ctx.debug_persist_counters()
order = np.argsort(ts)[::-1][:5]
print("calls %d (%s, save_weights %d): resident launches %d, abandoned %d, blocks re-joined %d, results differing from the first call's %d; median %.4f ms, mean %.4f ms, slowest %s ms at calls %s"
      % (n, arith, sw, l1 - l0, a1 - a0, r1 - r0, bad, 1e3 * np.median(ts), 1e3 * ts.mean(), [round(1e3 * ts[k], 3) for k in order], list(order)))
ctx.close()
