#!/usr/bin/env python3
"""The rounds of ONE resident launch (gn_fca_persist) on the device's own clock (100 MHz s_memrealtime), as eight of its blocks saw
them: loop top, level tables read + first record requested, records gathered, solved, pass (constants, taps used, pixel done,
returned), wave sums, block barrier, record stored. The stamps cost time themselves (~0.05-0.1 us each). Needs the stamps build:
  make -C egomotion_with_local_loop_closures_amd/csrc stamps ; ELLC_LIB_PATH=build/libellc_hip_stamps.so python3 tools/dbg/persist_trace.py [fast|exact]"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import diaglib  # noqa: F401
from egomotion_with_local_loop_closures_amd import api, synth, _lib
W, H, L = 640, 480, 4
NW = 12
fx, fy, cx, cy = synth.default_intrinsics(W, H)
arith = sys.argv[1] if len(sys.argv) > 1 else "fast"
p = synth.make_pair(W, H, seed=0x5EED)
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=1, max_frames=1, max_batch=1,
                                     arith=api.ARITH_FAST if arith == "fast" else api.ARITH_EXACT))
ctx.keyframe_upload(0, p["kf_image"]); ctx.keyframe_set_depth(0, p["depth0"], p["var0"]); ctx.frame_upload(0, p["cur_image"])
kf = np.zeros(1, np.int32)
for _ in range(60):
    pose, iters, _ = ctx.align(kf, kf)
buf = (C.c_ulonglong * (8 * 64 * NW))()
fn = _lib.lib().ellc_debug_persist_trace
fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
fn(ctx.h, buf)                 # (clears the trace)
N = 20
runs = []
for _ in range(N):
    pose, iters, _ = ctx.align(kf, kf)
    fn(ctx.h, buf)
    runs.append(np.array(list(buf), dtype=np.float64).reshape(8, 64, NW))
t = runs[-1]
subs = [0, 1, 8, 16, 48, 100, 200, 255]
t00 = t[0, 0, 0]
rounds = int((t[0, :, 0] > 0).sum())
print("%s: iterations %s, %d rounds; one launch, us since block 0's first loop top" % (arith, [int(v) for v in iters[0]], rounds))
print("round lvl | " + " | ".join("blk %3d: gathered  stored" % s for s in subs))
for r in range(rounds):
    lvl = int(t[0, r, 11]) >> 8
    def u(b, k):
        v = t[b, r, k]
        return "%6.2f" % ((v - t00) * 0.01) if v > 0 else "   -  "
    row = "%3d   %d  |" % (r, lvl)
    for b in range(8):
        row += "        %s %s%s |" % (u(b, 2), u(b, 10), "w" if int(t[b, r, 11]) & 1 else " ")
    print(row)
names = ["tables+1st rec", "gather wait", "solve", "consts", "taps used", "pixel", "pass tail", "wave sums", "barrier", "store", "loop end"]
for blk in (0, 2):
    print("block %d, median of %d launches, us per phase:" % (subs[blk], N))
    print("  round lvl  " + " ".join("%14s" % n for n in names) + "   round")
    for r in range(rounds):
        ph = []
        for run in runs:
            a = run[blk, r]
            nxt = run[blk, r + 1, 0] if r + 1 < rounds else 0.0
            seq = [a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], nxt]
            d = []
            prev = seq[0]
            for v in seq[1:]:
                if v > 0 and prev > 0:
                    d.append((v - prev) * 0.01); prev = v
                else:
                    d.append(0.0)
            d.append((nxt - a[0]) * 0.01 if nxt > 0 else 0.0)
            ph.append(d)
        m = np.median(np.array(ph), axis=0)
        print("  %3d    %d   " % (r, int(t[blk, r, 11]) >> 8) + " ".join("%14.2f" % v for v in m[:-1]) + "   %5.2f" % m[-1])
