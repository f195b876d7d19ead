"""Shared helpers for the parity tests: build the same problem in the oracle and in the HIP library."""
import numpy as np
from egomotion_with_local_loop_closures_amd import synth


def oracle_problem(O, w, h, levels, pair, early_exit=0, max_iter=(4, 7, 9, 12)):
    """Oracle-side keyframe / current frame / depth pyramid for one synthetic pair."""
    fx, fy, cx, cy = pair["intrinsics"]
    cfg = O.make_config(w, h, levels, fx, fy, cx, cy, max_iter=max_iter, early_exit=early_exit)
    kf = O.Frame(cfg, pair["kf_image"], 1)
    cur = O.Frame(cfg, pair["cur_image"], 2)
    dm = O.DepthMap(cfg)
    dm.set_keyframe(kf)
    deptharr0 = np.where(pair["depth0"] > 0, pair["depth0"], -1).astype(np.float32)
    dm.set_pyr0(deptharr0, pair["var0"])
    dm.build_inv_var_depth()
    kf.set_depth(0, pair["depth0"])
    dm.map_depth_to_keyframe()
    return cfg, kf, cur, dm


def gpu_problem(E, w, h, levels, pairs, early_exit=0, max_iter=(4, 7, 9, 12), diag=False, **kw):
    """HIP-side context with pairs[i] resident in keyframe slot i / frame slot i."""
    fx, fy, cx, cy = pairs[0]["intrinsics"]
    n = len(pairs)
    cfg = E.default_config(w, h, levels, fx=fx, fy=fy, cx=cx, cy=cy, early_exit=early_exit, max_iter=max_iter,
                           max_keyframes=max(n, kw.pop("max_keyframes", 1)), max_frames=max(n, kw.pop("max_frames", 1)),
                           max_batch=max(n, kw.pop("max_batch", 1)), **kw)
    ctx = E.Context(cfg, diag=diag)
    for i, p in enumerate(pairs):
        ctx.keyframe_upload(i, p["kf_image"])
        ctx.keyframe_set_depth(i, p["depth0"], p["var0"])
        ctx.frame_upload(i, p["cur_image"])
    return ctx


def bits_equal(a, b):
    """bit-exact f32 comparison that treats +0/-0 as equal and NaN==NaN."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return np.array_equal(a, b, equal_nan=True)
