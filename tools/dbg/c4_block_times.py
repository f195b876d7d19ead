#!/usr/bin/env python3
"""r06: how long the blocks of ONE C4 level-0 launch (gn_fca_dense4, 1280x960 dense x 64 alignments, 12 blocks each) live, on the device's
clock (stamps build: make -C .../csrc stamps; ELLC_LIB_PATH=build/libellc_hip_stamps.so python3 tools/dbg/c4_block_times.py)."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import diaglib  # noqa: F401
from egomotion_with_local_loop_closures_amd import api, synth, _lib
W, H, L, B = 1280, 960, 5, 16
fx, fy, cx, cy = synth.default_intrinsics(W, H)
pairs = [synth.make_pair(W, H, seed=0xC4 + i, dense=True) for i in range(4)]
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_iter=(4, 7, 9, 12, 12), max_keyframes=4 * B, max_frames=4 * B,
                                     max_batch=B, coalesce=4, concurrent_batches=16, arith=api.ARITH_FAST))
for b in range(4 * B):
    p = pairs[b % 4]
    ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"]); ctx.frame_upload(b, p["cur_image"])
slots = np.arange(4 * B, dtype=np.int32)
for r in range(3):
    ms, _, _ = ctx.profile_gn_kernel(slots, slots, 0, reps=3)
nb = 768
bs = (C.c_ulonglong * (4 * nb))()
_lib.lib().ellc_debug_block_stamps(ctx.h, bs, nb)
t = np.array(list(bs), dtype=np.float64).reshape(nb, 4) * 0.01
t0 = t[:, 0].min()
print("launch avg %.1f us (HIP events); per block, us since the first block's start:" % (ms * 1e3))
for name, col in (("start", 0), ("prologue end", 1), ("pixels done", 2), ("end", 3)):
    v = t[:, col] - t0
    print("  %-13s min %7.1f  p10 %7.1f  median %7.1f  p90 %7.1f  max %7.1f" % (name, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
dur = t[:, 2] - t[:, 1]
print("  pixel phase: min %.1f  median %.1f  max %.1f us" % (dur.min(), np.median(dur), dur.max()))
bid = np.arange(nb)
print("  pixel phase by chunk of the alignment (block index %% 12):", np.round([dur[(bid % 12) == k].mean() for k in range(12)], 1))
print("  by XCD (linear id %% 8):", np.round([dur[(bid % 8) == k].mean() for k in range(8)], 1))
al = bid // 12
print("  by scene (alignment %% 4):", np.round([dur[(al % 4) == k].mean() for k in range(4)], 1), " min/max per scene:", [(round(dur[(al % 4) == k].min(), 0), round(dur[(al % 4) == k].max(), 0)) for k in range(4)])
print("  by alignment // 8:", np.round([dur[(al // 8) == k].mean() for k in range(8)], 1))
third = bid // 256
print("  by dispatch round (block id // 256): pixel phase", np.round([dur[third == k].mean() for k in range(3)], 1), " end", np.round([(t[third == k, 3] - t0).mean() for k in range(3)], 1),
      " latest end", np.round([(t[third == k, 3] - t0).max() for k in range(3)], 1))
h, edges = np.histogram(dur, bins=12)
print("  histogram of the pixel phase:", list(zip(np.round(edges[:-1], 0), h)))
order = np.argsort(t[:, 3])
print("  last 10 blocks to end:", order[-10:], np.round(t[order[-10:], 3] - t0, 1))
