#!/usr/bin/env python3
"""How much of the bench loop's wall time the host spends inside ellc_align_enqueue (staging + graph launch) and inside
ellc_align_fetch (waiting for the device + copying the result): if the enqueue share approaches 1 the pipeline is host-bound.
usage: python3 tools/host_share.py [--steps N] [--mode fca|ica] [--arith fast|exact]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402

sys.argv = [sys.argv[0]] + [x for x in sys.argv[1:]]
a = bench.parse()
scenes = synth.make_shared_frame_batch(a.width, a.height, a.batch, seed=0x5EED)
wl = bench.Workload(api, a, scenes, a.arith, 0)
ctx, G, B = wl.ctx, wl.G, wl.B
wl.run(200)
n = a.steps
te = tf = 0.0
emax = 0.0
ctx.sync()
t0 = time.perf_counter()
for s in range(min(G, n)):
    t = time.perf_counter(); ctx.align_enqueue(wl.kf[s % G], wl.fr[s % G], mode=wl.mode); d = time.perf_counter() - t; te += d; emax = max(emax, d)
for s in range(n):
    t = time.perf_counter(); ctx.align_fetch(B); tf += time.perf_counter() - t
    if s + G < n:
        t = time.perf_counter(); ctx.align_enqueue(wl.kf[(s + G) % G], wl.fr[(s + G) % G], mode=wl.mode); d = time.perf_counter() - t; te += d; emax = max(emax, d)
ctx.sync()
T = time.perf_counter() - t0
print("steps %d: %.4f ms per step; in enqueue %.1f %% (%.1f us per batch, max %.1f us), in fetch %.1f %%, other %.1f %%" % (n, 1e3 * T / n, 100 * te / T, 1e6 * te / n, 1e6 * emax, 100 * tf / T, 100 * (T - te - tf) / T))
wl.close()
