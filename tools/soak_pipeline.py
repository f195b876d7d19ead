#!/usr/bin/env python3
"""Soak test of the concurrent batch pipeline (usage: soak_pipeline.py [steps] [B] [coalesce]): thousands of pipelined batches over the slot groups, with uploads of the
same images interleaved at random (they must not change anything, but they exercise the cross-stream ordering); every
fetched table must equal, bit for bit, the one the same group produced on its first, synchronous, run."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32   # 1 or 2: the state-driven early-exit schedule (continuations included)
CO = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # cfg.coalesce: 2 or 3 runs groups of batches per launch sequence, 4 x CO in flight
W, H, L = 640, 480, 4
G = 3 if CO == 1 else 4 * CO
fx, fy, cx, cy = synth.default_intrinsics(W, H)
pairs = [synth.make_pair(W, H, seed=0x5EED + i) for i in range(5)] + [synth.make_pair(W, H, seed=22, rot=0.03, trans=0.08)]   # the last: > 20 iterations
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=G * B, max_frames=G * B,
                                     max_batch=B, concurrent_batches=G, coalesce=CO))
for b in range(G * B):
    p = pairs[(b * 7) % len(pairs)]
    ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"]); ctx.frame_upload(b, p["cur_image"])
group = [np.arange(B, dtype=np.int32) + g * B for g in range(G)]
ref = [ctx.align(q, q) for q in group]
rng = np.random.default_rng(1)
bad = 0
t0 = time.perf_counter()
for s in range(G):
    ctx.align_enqueue(group[s % G], group[s % G])
for s in range(STEPS):
    got = ctx.align_fetch(B)
    r = ref[s % G]
    if not all(np.array_equal(x, y) for x, y in zip(got, r)):
        bad += 1
    if rng.random() < 0.05:   # same pixels again into a slot of the group that has just been fetched (its next batch is not enqueued yet)
        b = int(group[s % G][rng.integers(0, B)])
        ctx.frame_upload(b, pairs[(b * 7) % len(pairs)]["cur_image"])
    if s + G < STEPS:
        ctx.align_enqueue(group[(s + G) % G], group[(s + G) % G])
    if s % 500 == 0:
        print("step", s, "mismatches", bad, flush=True)
dt = time.perf_counter() - t0
print("steps %d, mismatching batches %d, %.3f ms per batch" % (STEPS, bad, 1e3 * dt / STEPS))
ctx.close()
sys.exit(1 if bad else 0)
