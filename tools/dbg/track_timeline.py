#!/usr/bin/env python3
"""The tracked-frame loop through ellc_track_frame (one call per frame), for a kernel trace:
  rocprofv3 --kernel-trace [--hip-runtime-trace] --output-format csv -d /tmp/tl -o tl -- python3 tools/dbg/track_timeline.py
  python3 tools/dbg/track_timeline.py --report /tmp/tl   (per-frame timeline of the last frames: start offset, duration, gap)"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    # with --hip-runtime-trace: when the host made the launch call (same clock), by correlation id
    api = {}
    for g in glob.glob(os.path.join(sys.argv[2], "**", "*hip_api_trace.csv"), recursive=True):
        for r in csv.DictReader(open(g)):
            if "Launch" in r["Function"]:
                api[r["Correlation_Id"]] = int(r["Start_Timestamp"])
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], api.get(r.get("Correlation_Id", ""), None)) for r in csv.DictReader(open(f))),
                  key=lambda r: r[0])
    # a frame starts at its ingest copy
    starts = [i for i, r in enumerate(rows) if "ingest_copy_u8" in r[2]]
    for fi in starts[-4:-2]:
        nxt = starts[starts.index(fi) + 1]
        t0 = rows[fi][0]
        print("frame of %d launches, %.1f us to the next frame's first launch" % (nxt - fi, (rows[nxt][0] - t0) / 1e3))
        prev_end = t0
        for s, e, n, h in rows[fi:nxt]:
            print("  +%7.1f us  %6.1f us  gap %5.1f  launched by the host at %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
                                                                                   "+%7.1f" % ((h - t0) / 1e3) if h else "      ?", n.split("(")[0][:60]))
            prev_end = max(prev_end, e)
    sys.exit(0)
import diaglib  # noqa: E402,F401
from egomotion_with_local_loop_closures_amd import api, synth  # noqa: E402
W, H, L = 640, 480, 4
pair = synth.make_pair(W, H, seed=0x5EED)
fx, fy, cx, cy = pair["intrinsics"]
ctx = api.Context(api.default_config(W, H, L, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=1, max_keyframes=2, max_frames=2, device=0, arith=api.ARITH_FAST))
ctx.keyframe_upload(0, pair["kf_image"]); ctx.keyframe_set_depth(0, pair["depth0"], pair["var0"])
st = synth.make_depth_state(W, H, 9, pair["kf_image"], pair["idepth_true"])
ctx.depth_set_keyframe(0); ctx.depth_set_state(st)
for f in range(int(os.environ.get("FRAMES", "120"))):
    ctx.frame_upload(f & 1, pair["cur_image"])
    ctx.track_frame(f & 1, save_weights=True)
ctx.sync(); ctx.close()
