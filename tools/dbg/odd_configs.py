#!/usr/bin/env python3
"""Unusual (max_batch, cfg.coalesce, early_exit) combinations: enqueue 4 x coalesce batches, FCA and ICA, and compare every
fetched result with the same call made synchronously (identical bits expected)."""
import os
import sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from egomotion_with_local_loop_closures_amd import api, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)
from helpers import gpu_problem
w, h, L = 160, 120, 3
pairs = [synth.make_pair(w, h, seed=700 + i, rot=0.004, trans=0.012) for i in range(6)]
for MB, co, ee in ((1, 4, 0), (1, 4, 1), (3, 3, 0), (5, 2, 1), (2, 4, 1), (6, 4, 0)):
    ctx = gpu_problem(api, w, h, L, pairs, early_exit=ee, max_iter=(3, 4, 5), max_batch=MB, concurrent_batches=4 * co, coalesce=co)
    rng = np.random.default_rng(MB * 10 + co)
    batches = [rng.integers(0, 6, size=MB).astype(np.int32) for _ in range(4 * co)]
    frames = [rng.integers(0, 6, size=MB).astype(np.int32) for _ in range(4 * co)]
    for mode in (0, 1):
        if mode == 1:
            for s in range(6):
                for l in range(L):
                    ctx.keyframe_set_weights(s, l, np.full((h >> l, w >> l), 0.03, np.float32), 1)
        ref = [ctx.align(k, f, mode=mode) for k, f in zip(batches, frames)]
        for k, f in zip(batches, frames):
            ctx.align_enqueue(k, f, mode=mode)
        for i in range(len(batches)):
            got = ctx.align_fetch(MB)
            assert all(np.array_equal(x, y) for x, y in zip(got, ref[i])), (MB, co, ee, mode, i)
    ctx.close()
    print("ok", MB, co, ee, flush=True)
