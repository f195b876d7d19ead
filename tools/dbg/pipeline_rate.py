#!/usr/bin/env python3
"""The batch workload (bench.py's) for the build ELLC_LIB_PATH names: a hash of the poses of 6 batches (builds compared bit for bit),
the median of eleven 20-step windows and the best of three 400-step blocks. usage: pipeline_rate.py [fast|exact] [fca|ica]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import diaglib  # noqa: E402,F401
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
arith = sys.argv[1] if len(sys.argv) > 1 else "fast"
mode = sys.argv[2] if len(sys.argv) > 2 else "fca"
sys.argv = sys.argv[:1]
a = bench.parse()
a.mode = mode
scenes = synth.make_shared_frame_batch(a.width, a.height, a.batch, seed=0x5EED)
wl = bench.Workload(api, a, scenes, arith, 0, shared_frame=True, prime=[5, 20])
wl.run(200); wl.ctx.sync()
h = hashlib.sha256()
wl.run(6, on_fetch=lambda p, i, w: (h.update(p.tobytes()), h.update(i.tobytes()), h.update(w.tobytes())))
blocks = sorted(1e3 * wl.timed(20)[0] / 20 for _ in range(11))
sus = min(1e3 * wl.timed(400)[0] / 400 for _ in range(3))
name = os.path.basename(os.environ.get("ELLC_LIB_PATH", "tree")).replace("libellc_hip_", "").replace(".so", "")
print("%-8s %s %s: poses %s  20-step windows median %.4f ms/step, 400 steps %.4f ms/step" % (name, mode, arith, h.hexdigest()[:16], blocks[5], sus))
wl.close()
