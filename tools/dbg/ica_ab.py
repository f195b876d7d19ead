#!/usr/bin/env python3
"""The batch workload in ELLC_MODE_ICA for the build ELLC_LIB_PATH names: poses of 6 batches (a hash, to compare builds bit for bit)
and the pipeline's rate. usage: ELLC_LIB_PATH=... ica_ab.py [fast|exact]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import diaglib  # noqa: E402,F401
import numpy as np  # noqa: E402
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
arith = sys.argv[1] if len(sys.argv) > 1 else "fast"
sys.argv = sys.argv[:1]
a = bench.parse()
a.mode = "ica"
scenes = synth.make_shared_frame_batch(a.width, a.height, a.batch, seed=0x5EED)
wl = bench.Workload(api, a, scenes, arith, 0, shared_frame=True, prime=[5, 20])
wl.run(200); wl.ctx.sync()
h = hashlib.sha256()
wl.run(6, on_fetch=lambda p, i, w: (h.update(p.tobytes()), h.update(i.tobytes()), h.update(w.tobytes())))
blocks = sorted(1e3 * wl.timed(20)[0] / 20 for _ in range(11))
sus = min(1e3 * wl.timed(400)[0] / 400 for _ in range(3))
name = os.path.basename(os.environ.get("ELLC_LIB_PATH", "tree")).replace("libellc_hip_", "").replace(".so", "")
print("%-8s ica %s: poses %s  20-step blocks median %.4f ms/step, 400 steps %.4f ms/step" % (name, arith, h.hexdigest()[:16], blocks[5], sus))
wl.close()
