#!/usr/bin/env python3
"""Wall time of every fetch in a long pipelined run of the bench workload (three batches in flight): where do slow steps
sit? usage: python tools/dbg/step_times.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
if len(sys.argv) > 2 and sys.argv[2] == "torch":   # as bench.py runs: torch imported, its context created on the device
    import torch
    torch.cuda.synchronize()
sys.argv = [sys.argv[0]]
a = bench.parse()
scenes = synth.make_shared_frame_batch(a.width, a.height, a.batch, seed=0xE11C)
wl = bench.Workload(api, a, scenes, "fast", 0)
wl.run(6)
stamps = []
wl.run(N, on_fetch=lambda *_: stamps.append(time.perf_counter()))
d = np.diff(np.array(stamps)) * 1e3
print("steps %d: median %.4f ms, mean %.4f ms, p99 %.3f ms, max %.3f ms" % (N, np.median(d), d.mean(), np.percentile(d, 99), d.max()))
slow = np.nonzero(d > 4 * np.median(d))[0]
print("steps slower than 4x the median:", len(slow), "at", slow[:40].tolist())
print("their durations (ms):", np.round(d[slow][:40], 2).tolist())
print("time in slow steps: %.1f %% of the run" % (100 * d[slow].sum() / d.sum()))
big = np.nonzero(d > 2.0)[0]
print("steps over 2 ms:", big.tolist(), np.round(d[big], 1).tolist())
wl.close()
