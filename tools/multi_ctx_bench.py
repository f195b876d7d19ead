#!/usr/bin/env python3
"""Aggregate rate of several contexts (one HIP stream each) driven round-robin from one host thread: do the latency-bound
coarse iterations of one batch overlap the throughput-bound fine iterations of another when they sit on different streams?
usage: multi_ctx_bench.py [n_contexts] [in_flight_per_context] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)

NC = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 1
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
W, H, L, B = 640, 480, 4, 32
sched = [4, 7, 9, 12]
fx, fy, cx, cy = synth.default_intrinsics(W, H)
pairs = [synth.make_pair(W, H, seed=0x5EED + i) for i in range(4)]
slots = np.arange(B, dtype=np.int32)
ctxs = []
for k in range(NC):
    ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_iter=sched, max_keyframes=NF * B,
                                         max_frames=NF * B, max_batch=B, concurrent_batches=min(3, NC * NF)))
    for b in range(NF * B):
        p = pairs[(b + k) % len(pairs)]
        ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"]); ctx.frame_upload(b, p["cur_image"])
    ctxs.append(ctx)


def run(steps):
    """steps batches per context; every context keeps NF batches in flight"""
    pending = [0] * NC
    done = [0] * NC
    issued = [0] * NC
    while min(done) < steps:
        for k, ctx in enumerate(ctxs):
            while pending[k] < NF and issued[k] < steps:
                g = slots + (issued[k] % NF) * B   # batches in flight on one context use their own slot group
                ctx.align_enqueue(g, g); pending[k] += 1; issued[k] += 1
        for k, ctx in enumerate(ctxs):
            if pending[k] and (pending[k] == NF or issued[k] == steps):
                ctx.align_fetch(B); pending[k] -= 1; done[k] += 1


run(5)
for c in ctxs:
    c.sync()
t0 = time.perf_counter()
run(STEPS)
for c in ctxs:
    c.sync()
dt = time.perf_counter() - t0
n = NC * STEPS
print(json.dumps({"contexts": NC, "in_flight_per_context": NF, "batches": n, "ms_per_batch": 1e3 * dt / n,
                  "gn_iterations_per_s": n * B * sum(sched) / dt}))
for c in ctxs:
    c.close()
