#!/usr/bin/env python3
"""The tracked-frame loop of bench.py alone: fused call vs separate calls, with / without the loop-closure batch."""
import json, os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)
import bench  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
a = argparse.Namespace(arith=sys.argv[1] if len(sys.argv) > 1 else "fast")
print(json.dumps({k: v for k, v in bench.tracked_frame(api, synth, a, 0).items() if k != "workload"}))
if not os.environ.get("ELLC_TRACK_NO_LC"):   # (rocprofv3 crashes inside its HIP interception when two host threads launch at once: the traced runs skip this loop)
    print(json.dumps({k: v for k, v in bench.tracked_frame(api, synth, a, 0, with_lc=True).items() if k != "workload"}))
