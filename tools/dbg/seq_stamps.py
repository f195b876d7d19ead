#!/usr/bin/env python3
"""The 32 Gauss-Newton launches of ONE launch sequence on the device's own clock, without a profiler: entry and exit of block (0, 0)
of every gn_fca_fused launch (s_memrealtime, 100 MHz), so that idle time between dependent launches of a replayed hipGraph shows.
Needs the experiment build: make -C egomotion_with_local_loop_closures_amd/csrc variant NAME=seqstamps DEFS=-DELLC_SEQ_STAMPS VARDIR=variants
usage: ELLC_LIB_PATH=.../variants/libellc_hip_seqstamps.so python3 tools/dbg/seq_stamps.py [--batch 128]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from egomotion_with_local_loop_closures_amd import api, synth, _lib  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))  # noqa: E402
import diaglib  # noqa: E402,F401

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
W, H, L, B = 640, 480, 4, a.batch
fx, fy, cx, cy = synth.default_intrinsics(W, H)
scenes = synth.make_shared_frame_batch(W, H, 32, seed=0x5EED)
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=0, max_keyframes=B, max_frames=1, max_batch=B, arith=api.ARITH_FAST))
ctx.frame_upload(0, scenes[0]["cur_image"])
for b in range(B):
    p = scenes[b % 32]
    ctx.keyframe_upload(b, p["kf_image"]); ctx.keyframe_set_depth(b, p["depth0"], p["var0"])
kf = np.arange(B, dtype=np.int32); fr = np.zeros(B, np.int32)
sched = [12, 9, 7, 4]   # launches per level, coarse to fine
lib = _lib.lib()
lib.ellc_debug_seq_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for r in range(a.reps):
    ctx.align(kf, fr)
    st = (C.c_ulonglong * 64)()
    assert lib.ellc_debug_seq_stamps(ctx.h, st) == 0
    t = np.array(list(st), dtype=np.float64) * 0.01   # us
    s, e = t[:32], t[32:]
    if r < 2:
        continue   # graph capture / warm-up
    gaps = s[1:] - e[:-1]   # exit of block (0,0) of launch n -> entry of block (0,0) of launch n + 1 (includes the rest of launch n's blocks)
    iv = s[1:] - s[:-1]
    print("rep %d: sequence %.1f us from the first launch's entry to the last one's exit" % (r, e[31] - s[0]))
    print("  launch-to-launch intervals (us):", " ".join("%.1f" % x for x in iv))
    print("  block (0,0) exit -> next entry  :", " ".join("%.1f" % x for x in gaps))
ctx.close()
