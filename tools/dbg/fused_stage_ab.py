#!/usr/bin/env python3
"""HIP-event time of createKeyFrame's one-launch regularise + fill + regularise (ellc_profile_depth_stage 4) for the library
ELLC_LIB_PATH names (experiment builds that leave stages out: where the kernel's time goes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import diaglib  # noqa: E402,F401  (ELLC_LIB_PATH -> _lib.use_library)
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
W, H, L = 640, 480, 4
pair = synth.make_pair(W, H, seed=31, rot=0.006, trans=0.03)
fx, fy, cx, cy = pair["intrinsics"]
st = synth.make_depth_state(W, H, 9, pair["kf_image"], pair["idepth_true"])
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, max_keyframes=2, max_frames=1, device=0))
ctx.keyframe_upload(0, pair["kf_image"]); ctx.frame_upload(0, pair["cur_image"])
ctx.depth_set_keyframe(0); ctx.depth_set_state(st)
for stage in (0, 4):
    ctx.depth_set_state(st)
    best = min(ctx.profile_depth_stage(stage, 0, pair["xi_true"], reps=20) for _ in range(5))
    print("%s stage %d: %.2f us" % (os.environ.get("ELLC_LIB_PATH", "tree").split("_")[-1], stage, 1e3 * best))
