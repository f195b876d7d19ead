#!/usr/bin/env python3
"""r06: what the per-frame upload costs the tracked frame (ellc_track_frame, fast mode): the loop as bench.py runs it (upload, then the
call), the same without any upload, and with the NEXT frame's upload issued in front of this frame's call (three frame slots). The
upload's host part (16 us) hides behind the depth stages; its two kernels run beside the line stereo's walk and the fill + regularise
launch and cost them 8 us together (rocprofv3: walk 29.8 -> 34.2 us, fill + regularise + export 13.6 -> 17.3) whichever way the calls
are ordered: 0.178-0.179 / 0.170 / 0.177 ms per frame. usage (GPU box, repo root): python3 tools/dbg/hostbound.py"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
from egomotion_with_local_loop_closures_amd import api, synth
W, H, L = 640, 480, 4
pair = synth.make_pair(W, H, seed=0x5EED)
fx, fy, cx, cy = pair["intrinsics"]
def run(upload, n=400):
    ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=2, max_frames=3, arith=api.ARITH_FAST))
    ctx.keyframe_upload(0, pair["kf_image"]); ctx.keyframe_set_depth(0, pair["depth0"], pair["var0"])
    st = synth.make_depth_state(W, H, 9, pair["kf_image"], pair["idepth_true"])
    ctx.depth_set_keyframe(0); ctx.depth_set_state(st)
    ctx.frame_upload(0, pair["cur_image"]); ctx.frame_upload(1, pair["cur_image"]); ctx.frame_upload(2, pair["cur_image"])
    for f in range(201):
        if upload == 1: ctx.frame_upload(f & 1, pair["cur_image"])
        if upload == 2: ctx.frame_upload((f + 1) % 3, pair["cur_image"])
        ctx.track_frame((f & 1) if upload < 2 else f % 3, save_weights=True)
    ctx.sync(); t0 = time.perf_counter()
    th = 0.0
    for f in range(n):
        if upload == 1: ctx.frame_upload(f & 1, pair["cur_image"])
        if upload == 2: ctx.frame_upload((f + 1) % 3, pair["cur_image"])   # the NEXT frame's image in front of this frame's call (three slots): its kernels run beside the resident launch
        t1 = time.perf_counter()
        ctx.track_frame((f & 1) if upload < 2 else f % 3, save_weights=True)
        th += time.perf_counter() - t1
    ctx.sync(); dt = (time.perf_counter() - t0) / n
    ctx.close()
    return 1e3 * dt, 1e3 * th / n
for rep in range(2):
    a = run(1); b = run(0); c = run(2)
    print("upload then track: %.4f ms per frame (call %.4f)   no upload: %.4f (call %.4f)   the next frame's upload in front of this frame's call, three slots: %.4f (call %.4f)" % (a[0], a[1], b[0], b[1], c[0], c[1]))
