#!/usr/bin/env python3
"""r06: P processes track the same sequence on ONE GPU at the same time (ellc_track_frame per frame, resident launches competing for
the device) — every process must produce the solo run's poses, iteration counts and seeds figures bit for bit, whatever was abandoned
and continued on the way. usage: track_contention.py [processes] [frames] [fast|exact]     (worker: track_contention.py --worker out.npz frames arith)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def worker(out, frames, arith):
    import diaglib  # noqa: F401
    from egomotion_with_local_loop_closures_amd import api, synth
    W, H, L = 640, 480, 4
    pair = synth.make_pair(W, H, seed=31, rot=0.006, trans=0.03)
    fx, fy, cx, cy = pair["intrinsics"]
    st = synth.make_depth_state(W, H, 9, pair["kf_image"], pair["idepth_true"])
    ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=2, max_frames=2,
                                         arith=api.ARITH_FAST if arith == "fast" else api.ARITH_EXACT))
    ctx.keyframe_upload(0, pair["kf_image"]); ctx.depth_set_keyframe(0); ctx.depth_set_state(st); ctx.depth_update_depth_image()
    rows = []
    t0 = time.perf_counter()
    for f in range(frames):
        ctx.frame_upload(f & 1, pair["cur_image"])
        pose, it, wgt, seeds = ctx.track_frame(f & 1, save_weights=True)
        rows.append(np.concatenate([pose, it.astype(np.float32), [wgt, seeds]]))
    dt = time.perf_counter() - t0
    launches, abandoned, rejoined = ctx.debug_persist_counters()
    np.savez(out, rows=np.array(rows, np.float32), stats=np.array([launches, abandoned, rejoined, 1e3 * dt / frames]))
    ctx.close()


if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    worker(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    sys.exit(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
arith = sys.argv[3] if len(sys.argv) > 3 else "fast"
tmp = os.path.join(ROOT, "gpurun_out", "contention"); os.makedirs(tmp, exist_ok=True)
me = os.path.abspath(__file__)
subprocess.run([sys.executable, me, "--worker", os.path.join(tmp, "solo.npz"), str(frames), arith], check=True)
solo = np.load(os.path.join(tmp, "solo.npz"))
procs = [subprocess.Popen([sys.executable, me, "--worker", os.path.join(tmp, "p%d.npz" % i), str(frames), arith]) for i in range(P)]
rc = [p.wait() for p in procs]
print("solo: %d resident launches, %d abandoned, %.4f ms per frame" % (solo["stats"][0], solo["stats"][1], solo["stats"][3]))
ok = all(r == 0 for r in rc)
for i in range(P):
    d = np.load(os.path.join(tmp, "p%d.npz" % i))
    same = np.array_equal(d["rows"], solo["rows"])
    bad = int((~np.all(d["rows"] == solo["rows"], axis=1)).sum())
    print("process %d of %d: %d resident launches, %d abandoned, %d blocks re-joined, %.4f ms per frame, frames that differ from the solo run: %d" % (
        i, P, d["stats"][0], d["stats"][1], d["stats"][2], d["stats"][3], bad))
    ok = ok and same
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
