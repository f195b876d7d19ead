#!/usr/bin/env python3
"""The timed workload at the driver's 20 steps against batches in flight / batches per launch group: median of 15 blocks of 20 steps
(each bracketed by a sync like the timed region: the pipeline's fill and drain are inside) and one block of 400 steps."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import diaglib  # noqa: E402,F401
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.argv = sys.argv[:1]
a = bench.parse()
scenes = synth.make_shared_frame_batch(a.width, a.height, a.batch, seed=0x5EED)
# (r04, a build with ellc_ctx::MAX_COALESCE = 8: groups of 5 .. 8 batches are slower — 0.143 .. 0.171 ms/step at 20 steps, 0.128 .. 0.136 sustained)
for inflight, coalesce in ((16, 4), (12, 4), (8, 4), (12, 3), (8, 2), (16, 4)):
    a.inflight, a.coalesce = inflight, coalesce
    wl = bench.Workload(api, a, scenes, "fast", 0, shared_frame=True, prime=[5, 20])
    wl.run(200); wl.ctx.sync()
    ms = sorted(1e3 * wl.timed(20)[0] / 20 for _ in range(15))
    sus = 1e3 * wl.timed(400)[0] / 400
    print("in flight %2d (G %2d), %d per group: 20-step blocks median %.4f min %.4f ms/step; 400 steps %.4f ms/step" % (inflight, wl.G, coalesce, ms[7], ms[0], sus))
    wl.close()
