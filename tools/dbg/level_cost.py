#!/usr/bin/env python3
"""What each pyramid level costs the pipelined workload: the sustained ms per step with the level's iterations removed from the
schedule (cfg.max_iter; index = level, 0 the finest). An upper bound on what making that level's launches free would give."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import diaglib  # noqa: E402,F401
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.argv = sys.argv[:1]
a = bench.parse()
scenes = synth.make_shared_frame_batch(a.width, a.height, a.batch, seed=0x5EED)
for sched in ([4, 7, 9, 12], [4, 7, 9, 1], [4, 7, 1, 1], [4, 1, 1, 1], [1, 7, 9, 12], [4, 7, 9, 12]):
    wl = bench.Workload(api, a, scenes, "fast", 0, shared_frame=True, prime=[5, 20], sched=sched)
    wl.run(200); wl.ctx.sync()
    sus = min(1e3 * wl.timed(400)[0] / 400 for _ in range(3))
    print("schedule %-16s (%2d launches): %.4f ms per step sustained" % (sched, sum(sched), sus))
    wl.close()
