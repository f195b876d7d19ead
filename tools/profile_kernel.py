#!/usr/bin/env python3
"""Runs only the dominant kernel (gn_fca_fused, level 0; ellc_profile_gn_kernel) over a resident batch, plus the counter-calibration
kernel — the process rocprofv3 wraps when collecting --kernel-trace / --pmc summaries for profiles/."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # noqa: E402
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library: diagnostic builds)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--level", type=int, default=0)
ap.add_argument("--dense", action="store_true")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--calib-mb", type=int, default=512)
ap.add_argument("--inflight", type=int, default=16, help="cfg.concurrent_batches, as bench.py's default (it sizes the grid)")
ap.add_argument("--coalesce", type=int, default=4, help="cfg.coalesce, as bench.py's default: the launch covers this many batches side by side")
ap.add_argument("--arith", choices=["fast", "exact"], default="fast")
ap.add_argument("--device-warmup", type=int, default=300, help="untimed launches before the measured ones")
ap.add_argument("--rot", type=float, default=0.01, help="rotation of the synthetic camera motion (rad)")
ap.add_argument("--trans", type=float, default=0.02, help="translation of the synthetic camera motion")
a = ap.parse_args()
W, H, L, B = a.width, a.height, a.levels, a.batch
K = B * a.coalesce   # alignments of one launch: a group of full batches
fx, fy, cx, cy = synth.default_intrinsics(W, H)
# the bench workload: B keyframes of one scene against ONE frame (semi-dense); dense (C4 shape): two scenes, own frame each
pairs = [synth.make_pair(W, H, seed=0xC4 + i, dense=True, rot=a.rot, trans=a.trans) for i in range(2)] if a.dense else synth.make_shared_frame_batch(W, H, B, seed=0x5EED, rot=a.rot, trans=a.trans)
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_keyframes=K, max_frames=K, max_batch=B,
                                     concurrent_batches=a.inflight, coalesce=a.coalesce, arith=api.ARITH_FAST if a.arith == "fast" else api.ARITH_EXACT))
for b in range(K):
    p = pairs[b % len(pairs)]
    ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"])
    if a.dense or b % B == 0:
        ctx.frame_upload(b if a.dense else b // B, p["cur_image"])
slots = np.arange(K, dtype=np.int32)
fr_slots = slots if a.dense else (slots // B).astype(np.int32)
if a.device_warmup > 0:   # the same launches, untimed, until the device is in its working state (as bench.py's --device-warmup)
    ctx.profile_gn_kernel(slots, fr_slots, a.level, reps=a.device_warmup)
ms, alg, V = ctx.profile_gn_kernel(slots, fr_slots, a.level, reps=a.reps)
cal_bytes = a.calib_mb << 20
cms = ctx.profile_calibrate_read(cal_bytes, reps=5)
print(json.dumps({"kernel": "gn_fca_fused", "arith": a.arith, "size": [W, H, L], "dense": bool(a.dense), "level": a.level, "batch": B, "coalesce": a.coalesce, "alignments_per_launch": K, "concurrent_batches": a.inflight, "avg_ms": ms, "algorithmic_bytes": alg, "valid_pixels": V,
                  "achieved_GBps": alg / ms / 1e6, "launches": a.reps + 3, "warmup_launches": (a.device_warmup + 3) if a.device_warmup > 0 else 0, "calib_bytes_per_launch": cal_bytes, "calib_avg_ms": cms,
                  "calib_GBps": cal_bytes / cms / 1e6, "calib_launches": 6}))
ctx.close()
