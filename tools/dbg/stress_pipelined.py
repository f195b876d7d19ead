#!/usr/bin/env python3
"""One-off wide run of tests/test_gpu_gn.py::test_pipelined_calls_equal_the_same_calls_made_one_by_one over many seeds and
every cfg.coalesce (random mixes of full / partial batches, modes, saved weights, uploads, depth updates, fetches; pipelined
context against a synchronous one, results must be identical). usage: stress_pipelined.py [first_seed] [n_seeds] [fast] [cache]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from egomotion_with_local_loop_closures_amd import api  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)
import test_gpu_gn  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
CACHE = 1 if "cache" in sys.argv[3:] else 0   # the pipelined context keeps the compact lists with the slots (cfg.cache_records)
if "fast" in sys.argv[3:]:   # the same sequences in the tolerance arithmetic mode (both contexts)
    _gp = test_gpu_gn.gpu_problem
    test_gpu_gn.gpu_problem = lambda *a, **kw: _gp(*a, **dict(kw, arith=api.ARITH_FAST))
bad = 0
for seed in range(first, first + n):
    for coalesce, concurrent in ((1, 3), (2, 8), (3, 12), (4, 16)):
        try:
            test_gpu_gn.test_pipelined_calls_equal_the_same_calls_made_one_by_one(api, seed, concurrent, coalesce, cache=CACHE)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("seed %d coalesce %d: %r" % (seed, coalesce, e), flush=True)
    if seed % 5 == 0:
        print("seed", seed, "failures so far", bad, flush=True)
print("done: %d seeds x 4 configurations, %d failures" % (n, bad))
sys.exit(1 if bad else 0)
