"""Thin Python view of the C ABI (numpy in / numpy out). All compute happens in libellc_hip.so on the GPU."""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import EllcConfig, EllcHypotheses, EllcError, MAX_LEVELS

MODE_FCA = 0
MODE_ICA = 1
ARITH_EXACT = 0
ARITH_FAST = 1
HYP_FIELDS = ("invDepth", "invDepthSmoothed", "variance", "varianceSmoothed", "validity", "blacklisted", "valid")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_config(width, height, levels=4, **kw):
    cfg = EllcConfig()
    _lib.lib().ellc_default_config(C.byref(cfg), width, height, levels)
    for k, v in kw.items():
        if k == "max_iter":
            mi = list(v) + [12] * (MAX_LEVELS - len(v))
            for i in range(MAX_LEVELS):
                cfg.max_iter[i] = mi[i]
        else:
            setattr(cfg, k, v)
    return cfg


default_diag = False   # tools/diaglib.py sets it: tools create their contexts in the diagnostic library


class Context:
    """Resident keyframe / frame slots, one depth map and an in-order queue of work on the GPU (ellc_ctx); up to three
    alignment batches may be in flight at once (align_enqueue / align_fetch) — 4 x cfg.coalesce with cfg.coalesce > 1, where
    full batches enqueued one after the other run side by side in one launch sequence."""

    def __init__(self, cfg, diag=None):
        """diag=True: the context lives in libellc_hip_diag.so (include/ellc_abi_diag.h: profile_* / selftest_* / debug_* below) — the
        same kernels and launch paths as the shipping library, which does not export those hooks."""
        self.cfg = cfg
        self.diag = default_diag if diag is None else bool(diag)
        self._l = _lib.diag_lib() if self.diag else _lib.lib()
        h = C.c_void_p()
        st = self._l.ellc_ctx_create(C.byref(cfg), C.byref(h))
        self.h = h
        if st != 0:
            msg = self._l.ellc_last_error(h).decode() if h else "context creation failed"
            if h:
                self._l.ellc_ctx_destroy(h)
                self.h = None
            raise EllcError("ellc_ctx_create -> %d: %s" % (st, msg))
        self.levels = cfg.levels

    def close(self):
        if getattr(self, "h", None):
            self._l.ellc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, what):
        if st != 0:
            raise EllcError("%s -> %d: %s" % (what, st, self._l.ellc_last_error(self.h).decode()))

    def sync(self):
        self._ck(self._l.ellc_sync(self.h), "ellc_sync")

    def counters(self):
        """Diagnostic counters (ELLC_CTR_* of ellc_abi.h): polled / poll_timeout / event_wait / continuation."""
        out = (C.c_longlong * 4)()
        self._ck(self._l.ellc_ctx_counters(self.h, out, 4), "ellc_ctx_counters")
        return dict(polled=out[0], poll_timeout=out[1], event_wait=out[2], continuation=out[3])

    def set_poll_timeout_us(self, us):
        self._ck(self._l.ellc_ctx_set_poll_timeout_us(self.h, int(us)), "ellc_ctx_set_poll_timeout_us")

    def set_grid_batch(self, n):
        """cfg.grid_batch for the calls that follow (ellc_ctx_set_grid_batch)."""
        self._ck(self._l.ellc_ctx_set_grid_batch(self.h, int(n)), "ellc_ctx_set_grid_batch")

    def set_dense_maps(self, mode):
        """0: dense maps take the list-free kernels (default); 1: always the compact lists (ellc_ctx_set_dense_maps)."""
        self._ck(self._l.ellc_ctx_set_dense_maps(self.h, int(mode)), "ellc_ctx_set_dense_maps")

    def set_persistent_schedule(self, mode):
        """1: the state-driven schedule as one resident launch (default); 0: one launch per iteration; 2: test hook, every resident
        launch is abandoned at its first hand-over (ellc_ctx_set_persistent_schedule)."""
        self._ck(self._l.ellc_ctx_set_persistent_schedule(self.h, int(mode)), "ellc_ctx_set_persistent_schedule")

    def level_shape(self, level):
        return (self.cfg.height >> level, self.cfg.width >> level)

    # ---- frame side
    def frame_upload(self, slot, image):
        image = np.ascontiguousarray(image, np.uint8)
        assert image.shape == (self.cfg.height, self.cfg.width)
        self._ck(self._l.ellc_frame_upload(self.h, slot, _p(image)), "ellc_frame_upload")

    def keyframe_upload(self, slot, image):
        image = np.ascontiguousarray(image, np.uint8)
        assert image.shape == (self.cfg.height, self.cfg.width)
        self._ck(self._l.ellc_keyframe_upload(self.h, slot, _p(image)), "ellc_keyframe_upload")

    def keyframe_from_frame(self, kf_slot, frame_slot):
        self._ck(self._l.ellc_keyframe_from_frame(self.h, kf_slot, frame_slot), "ellc_keyframe_from_frame")

    def image_level(self, is_kf, slot, level):
        sw, sh, c, r = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._ck(self._l.ellc_get_image_level(self.h, int(is_kf), slot, level, None, C.byref(sw), C.byref(sh), C.byref(c), C.byref(r)),
                 "ellc_get_image_level")
        out = np.zeros((sh.value, sw.value), np.uint8)
        self._ck(self._l.ellc_get_image_level(self.h, int(is_kf), slot, level, _p(out), None, None, None, None), "ellc_get_image_level")
        return out, (r.value, c.value)

    def gradient(self, is_kf, slot, level):
        shp = self.level_shape(level)
        gx = np.zeros(shp, np.float32); gy = np.zeros(shp, np.float32)
        self._ck(self._l.ellc_get_gradient(self.h, int(is_kf), slot, level, _p(gx), _p(gy)), "ellc_get_gradient")
        return gx, gy

    def max_gradient(self, is_kf, slot):
        out = np.zeros((self.cfg.height, self.cfg.width), np.float32)
        n = C.c_int(0)
        self._ck(self._l.ellc_get_max_gradient(self.h, int(is_kf), slot, _p(out), C.byref(n)), "ellc_get_max_gradient")
        return out, n.value

    # ---- keyframe depth / weights
    def keyframe_set_depth(self, slot, depth0, var0):
        d = np.ascontiguousarray(depth0, np.float32); v = np.ascontiguousarray(var0, np.float32)
        self._ck(self._l.ellc_keyframe_set_depth(self.h, slot, _p(d), _p(v)), "ellc_keyframe_set_depth")

    def keyframe_set_depth_level(self, slot, level, depth, var):
        d = np.ascontiguousarray(depth, np.float32); v = np.ascontiguousarray(var, np.float32)
        self._ck(self._l.ellc_keyframe_set_depth_level(self.h, slot, level, _p(d), _p(v)), "ellc_keyframe_set_depth_level")

    def keyframe_depth_level(self, slot, level):
        shp = self.level_shape(level)
        d = np.zeros(shp, np.float32); v = np.zeros(shp, np.float32)
        self._ck(self._l.ellc_keyframe_get_depth_level(self.h, slot, level, _p(d), _p(v)), "ellc_keyframe_get_depth_level")
        return d, v

    def keyframe_set_weights(self, slot, level, w, num_added=1):
        w = np.ascontiguousarray(w, np.float32)
        self._ck(self._l.ellc_keyframe_set_weights(self.h, slot, level, _p(w), num_added), "ellc_keyframe_set_weights")

    def keyframe_weights(self, slot, level):
        w = np.zeros(self.level_shape(level), np.float32)
        n = C.c_int(0)
        self._ck(self._l.ellc_keyframe_get_weights(self.h, slot, level, _p(w), C.byref(n)), "ellc_keyframe_get_weights")
        return w, n.value

    def keyframe_finalise_weights(self, slot):
        self._ck(self._l.ellc_keyframe_finalise_weights(self.h, slot), "ellc_keyframe_finalise_weights")

    # ---- alignment
    def _batch(self, kf_slots, frame_slots, init_pose):
        kf = np.ascontiguousarray(kf_slots, np.int32).reshape(-1)
        fr = np.ascontiguousarray(frame_slots, np.int32).reshape(-1)
        B = kf.size
        assert fr.size == B
        ip = np.zeros((B, 6), np.float32) if init_pose is None else np.ascontiguousarray(init_pose, np.float32).reshape(B, 6)
        return B, kf, fr, ip

    def align(self, kf_slots, frame_slots, init_pose=None, mode=MODE_FCA, save_weights=False):
        B, kf, fr, ip = self._batch(kf_slots, frame_slots, init_pose)
        pose = np.zeros((B, 6), np.float32); iters = np.zeros((B, self.levels), np.int32); wgt = np.zeros(B, np.float32)
        self._ck(self._l.ellc_align(self.h, B, _p(kf), _p(fr), _p(ip), mode, int(save_weights), _p(pose), _p(iters), _p(wgt)), "ellc_align")
        return pose, iters, wgt

    def align_enqueue(self, kf_slots, frame_slots, init_pose=None, mode=MODE_FCA, save_weights=False):
        B, kf, fr, ip = self._batch(kf_slots, frame_slots, init_pose)
        self._ck(self._l.ellc_align_enqueue(self.h, B, _p(kf), _p(fr), _p(ip), mode, int(save_weights)), "ellc_align_enqueue")
        return B

    def align_fetch(self, B):
        pose = np.zeros((B, 6), np.float32); iters = np.zeros((B, self.levels), np.int32); wgt = np.zeros(B, np.float32)
        self._ck(self._l.ellc_align_fetch(self.h, B, _p(pose), _p(iters), _p(wgt)), "ellc_align_fetch")
        return pose, iters, wgt

    def gn_iterate(self, kf_slot, frame_slot, level, pose, mode=MODE_FCA, it=0, planes=False):
        pose = np.ascontiguousarray(pose, np.float32)
        H = np.zeros((6, 6), np.float32); b = np.zeros(6, np.float32); d = np.zeros(6, np.float32); p = np.zeros(6, np.float32)
        w = C.c_float(0)
        shp = self.level_shape(level)
        pl = np.zeros((10,) + shp, np.float32) if planes else None
        self._ck(self._l.ellc_gn_iterate(self.h, kf_slot, frame_slot, level, mode, it, _p(pose), _p(H), _p(b), _p(d), _p(p), C.byref(w), _p(pl)),
                 "ellc_gn_iterate")
        out = dict(H=H, b=b, delta=d, pose=p, weighted=w.value)
        if planes:
            out.update(residual=pl[0], weight=pl[1], warpedX=pl[2], warpedY=pl[3], J=pl[4:10])
        return out

    def gn_display_planes(self, kf_slot, frame_slot, level, pose):
        """display_templateimg / display_2bewarpedimg (u8), display_warpedimg / display_origres (f32) of one pass."""
        pose = np.ascontiguousarray(pose, np.float32)
        shp = self.level_shape(level)
        t = np.zeros(shp, np.uint8); k = np.zeros(shp, np.uint8); w = np.zeros(shp, np.float32); o = np.zeros(shp, np.float32)
        self._ck(self._l.ellc_gn_display_planes(self.h, kf_slot, frame_slot, level, _p(pose), _p(t), _p(k), _p(w), _p(o)), "ellc_gn_display_planes")
        return dict(templateimg=t, tobewarpedimg=k, warpedimg=w, origres=o)

    # ---- depth map
    def _hyp(self, st):
        arrs = [np.ascontiguousarray(st["invDepth"], np.float32), np.ascontiguousarray(st["invDepthSmoothed"], np.float32),
                np.ascontiguousarray(st["variance"], np.float32), np.ascontiguousarray(st["varianceSmoothed"], np.float32),
                np.ascontiguousarray(st["validity"], np.int32), np.ascontiguousarray(st["blacklisted"], np.int32),
                np.ascontiguousarray(st["valid"], np.uint8)]
        h = EllcHypotheses(*[a.ctypes.data for a in arrs])
        return h, arrs

    def depth_set_state(self, st):
        h, keep = self._hyp(st)
        self._ck(self._l.ellc_depth_set_state(self.h, C.byref(h)), "ellc_depth_set_state")

    def depth_get_state(self):
        shp = (self.cfg.height, self.cfg.width)
        st = dict(invDepth=np.zeros(shp, np.float32), invDepthSmoothed=np.zeros(shp, np.float32), variance=np.zeros(shp, np.float32),
                  varianceSmoothed=np.zeros(shp, np.float32), validity=np.zeros(shp, np.int32), blacklisted=np.zeros(shp, np.int32),
                  valid=np.zeros(shp, np.uint8))
        h, keep = self._hyp(st)
        self._ck(self._l.ellc_depth_get_state(self.h, C.byref(h)), "ellc_depth_get_state")
        return dict(zip(HYP_FIELDS, keep))

    def depth_set_keyframe(self, slot):
        self._ck(self._l.ellc_depth_set_keyframe(self.h, slot), "ellc_depth_set_keyframe")

    def depth_propagate(self, new_kf_slot, pose_new_wrt_old):
        p = np.ascontiguousarray(pose_new_wrt_old, np.float32)
        self._ck(self._l.ellc_depth_propagate(self.h, new_kf_slot, _p(p)), "ellc_depth_propagate")

    def depth_observe(self, frame_slot, pose_frame_wrt_kf):
        p = np.ascontiguousarray(pose_frame_wrt_kf, np.float32)
        self._ck(self._l.ellc_depth_observe(self.h, frame_slot, _p(p)), "ellc_depth_observe")

    def depth_fill_holes(self):
        self._ck(self._l.ellc_depth_fill_holes(self.h), "ellc_depth_fill_holes")

    def depth_regularize(self, remove_occlusions=False):
        self._ck(self._l.ellc_depth_regularize(self.h, int(remove_occlusions)), "ellc_depth_regularize")

    def depth_do_regularization(self, remove_occlusions=False):
        """doRegularization (:1627-1635): fill holes + regularise in one launch."""
        self._ck(self._l.ellc_depth_do_regularization(self.h, int(remove_occlusions)), "ellc_depth_do_regularization")

    def depth_regularize_fill_regularize(self, remove_occlusions=True):
        """regularise(remove_occlusions) + fill holes + regularise(False) in one launch (createKeyFrame's middle)"""
        self._ck(self._l.ellc_depth_regularize_fill_regularize(self.h, int(remove_occlusions)), "ellc_depth_regularize_fill_regularize")

    def depth_make_inv_depth_one(self):
        f = C.c_float(0)
        self._ck(self._l.ellc_depth_make_inv_depth_one(self.h, C.byref(f)), "ellc_depth_make_inv_depth_one")
        return f.value

    def depth_update_depth_image(self):
        self._ck(self._l.ellc_depth_update_depth_image(self.h), "ellc_depth_update_depth_image")

    def depth_create_keyframe(self, new_kf_slot, pose_new_wrt_old):
        p = np.ascontiguousarray(pose_new_wrt_old, np.float32)
        f = C.c_float(0)
        self._ck(self._l.ellc_depth_create_keyframe(self.h, new_kf_slot, _p(p), C.byref(f)), "ellc_depth_create_keyframe")
        return f.value

    def track_frame(self, frame_slot, init_pose=None, save_weights=False):
        """ellc_track_frame: alignment against the depth map's keyframe + observe / fill holes / regularise / export behind it on the
        device. Returns pose (6,), iters (levels,), weightedPose, seeds percent (of the map before the observation)."""
        ip = np.zeros(6, np.float32) if init_pose is None else np.ascontiguousarray(init_pose, np.float32).reshape(6)
        pose = np.zeros(6, np.float32); iters = np.zeros(self.levels, np.int32); w = C.c_float(0); sd = C.c_float(0)
        self._ck(self._l.ellc_track_frame(self.h, frame_slot, _p(ip), int(save_weights), _p(pose), _p(iters), C.byref(w), C.byref(sd)), "ellc_track_frame")
        return pose, iters, w.value, sd.value

    def depth_seeds(self):
        f = C.c_float(0)
        self._ck(self._l.ellc_depth_seeds(self.h, C.byref(f)), "ellc_depth_seeds")
        return f.value

    # ---- frame ingest (Frame.cpp:45-75 after the decode)
    def ingest_configure(self, orig_w, orig_h, fx, fy, cx, cy, dist5=None, do_undistort=True):
        d = np.ascontiguousarray(dist5 if dist5 is not None else np.zeros(5), np.float32)
        kn = np.zeros(4, np.float32)
        self._ck(self._l.ellc_ingest_configure(self.h, int(orig_w), int(orig_h), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                               _p(d), int(bool(do_undistort)), _p(kn)), "ellc_ingest_configure")
        return kn

    def frame_ingest_bgr(self, slot, bgr, probes=False):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        W, H = self.cfg.width, self.cfg.height
        if probes:
            g = np.zeros((H, W), np.uint8); u = np.zeros((H, W, 4), np.uint8)
            self._ck(self._l.ellc_frame_ingest_bgr(self.h, slot, _p(bgr), _p(g), _p(u)), "ellc_frame_ingest_bgr")
            return g, u
        self._ck(self._l.ellc_frame_ingest_bgr(self.h, slot, _p(bgr), None, None), "ellc_frame_ingest_bgr")

    # ---- loop-closure support
    def histogram(self, is_kf, slot):
        h = np.zeros(256, np.float32)
        self._ck(self._l.ellc_histogram(self.h, int(is_kf), slot, _p(h)), "ellc_histogram")
        return h

    def copy_slot(self, dst_is_kf, dst, src_is_kf, src):
        self._ck(self._l.ellc_copy_slot(self.h, int(dst_is_kf), dst, int(src_is_kf), src), "ellc_copy_slot")

    # ---- measurement hooks, self-tests, test hooks: contexts created with diag=True only (include/ellc_abi_diag.h)
    def _need_diag(self, what):
        if not self.diag and _lib.DIAG_SO_PATH != _lib.SO_PATH:
            raise EllcError("%s is a diagnostic entry point (include/ellc_abi_diag.h): create the context with diag=True" % what)

    def debug_persist_delay(self, first_block, polls):
        self._need_diag("ellc_debug_persist_delay")
        self._ck(self._l.ellc_debug_persist_delay(self.h, int(first_block), int(polls)), "ellc_debug_persist_delay")

    def debug_set_persist_epoch(self, epoch):
        self._need_diag("ellc_debug_set_persist_epoch")
        self._ck(self._l.ellc_debug_set_persist_epoch(self.h, C.c_uint(epoch)), "ellc_debug_set_persist_epoch")

    def debug_set_eager_lists(self, on):
        self._need_diag("ellc_debug_set_eager_lists")
        self._ck(self._l.ellc_debug_set_eager_lists(self.h, int(bool(on))), "ellc_debug_set_eager_lists")

    def debug_set_fold_staging(self, on):
        self._need_diag("ellc_debug_set_fold_staging")
        self._ck(self._l.ellc_debug_set_fold_staging(self.h, int(bool(on))), "ellc_debug_set_fold_staging")

    def debug_set_hinv_cache(self, on):
        self._need_diag("ellc_debug_set_hinv_cache")
        self._ck(self._l.ellc_debug_set_hinv_cache(self.h, int(bool(on))), "ellc_debug_set_hinv_cache")

    def debug_persist_counters(self):
        self._need_diag("ellc_debug_persist_counters")
        a = C.c_longlong(0); b = C.c_longlong(0); r = C.c_longlong(0)
        self._ck(self._l.ellc_debug_persist_counters(self.h, C.byref(a), C.byref(b), C.byref(r)), "ellc_debug_persist_counters")
        return a.value, b.value, r.value

    def profile_gn_kernel(self, kf_slots, frame_slots, level, reps=20):
        self._need_diag("ellc_profile_gn_kernel")
        B, kf, fr, _ = self._batch(kf_slots, frame_slots, None)
        ms = C.c_float(0); by = C.c_double(0); v = C.c_longlong(0)
        self._ck(self._l.ellc_profile_gn_kernel(self.h, B, _p(kf), _p(fr), level, reps, C.byref(ms), C.byref(by), C.byref(v)),
                 "ellc_profile_gn_kernel")
        return ms.value, by.value, v.value

    def selftest_div_pair(self, a, b):
        self._need_diag("ellc_selftest_div_pair")
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        qp = np.zeros_like(a); qr = np.zeros_like(a)
        self._ck(self._l.ellc_selftest_div_pair(self.h, int(a.size), _p(a), _p(b), _p(qp), _p(qr)), "ellc_selftest_div_pair")
        return qp, qr

    def selftest_lu(self, tri21):
        self._need_diag("ellc_selftest_lu")
        tri21 = np.ascontiguousarray(tri21, np.float64).reshape(-1, 21)
        out = np.zeros((tri21.shape[0], 6, 6), np.float32)
        self._ck(self._l.ellc_selftest_lu(self.h, int(tri21.shape[0]), _p(tri21), _p(out)), "ellc_selftest_lu")
        return out

    def profile_calibrate_read(self, nbytes, reps=10):
        self._need_diag("ellc_profile_calibrate_read")
        ms = C.c_float(0)
        self._ck(self._l.ellc_profile_calibrate_read(self.h, C.c_size_t(nbytes), reps, C.byref(ms)), "ellc_profile_calibrate_read")
        return ms.value

    def profile_depth_stage(self, stage, frame_slot, pose, reps=20):
        """ms per call of one depth-map stage (0 regularize, 1 fill holes, 2 observe, 3 update depth image, 4 createKeyFrame's
        regularise + fill + regularise in one launch, 5 the tracked frame's fill + regularise + update depth image in one launch),
        HIP events."""
        self._need_diag("ellc_profile_depth_stage")
        pose = np.ascontiguousarray(pose, np.float32)
        ms = C.c_float(0)
        self._ck(self._l.ellc_profile_depth_stage(self.h, stage, frame_slot, _p(pose), reps, C.byref(ms)), "ellc_profile_depth_stage")
        return ms.value

    def profile_stream_read(self, nbytes, reps=10):
        self._need_diag("ellc_profile_stream_read")
        ms = C.c_float(0)
        self._ck(self._l.ellc_profile_stream_read(self.h, C.c_size_t(nbytes), reps, C.byref(ms)), "ellc_profile_stream_read")
        return ms.value

    def profile_align(self, kf_slots, frame_slots, init_pose=None, mode=MODE_FCA, reps=5):
        self._need_diag("ellc_profile_align")
        B, kf, fr, ip = self._batch(kf_slots, frame_slots, init_pose)
        ms = C.c_float(0)
        self._ck(self._l.ellc_profile_align(self.h, B, _p(kf), _p(fr), _p(ip), mode, reps, C.byref(ms)), "ellc_profile_align")
        return ms.value


def copy_slot_across(dst_ctx, dst_is_kf, dst, src_ctx, src_is_kf, src):
    """ellc_copy_slot_across: a slot's planes from one context into another on the same device (the loop-closure ring's deep copy)."""
    if dst_ctx._l is not src_ctx._l:
        raise EllcError("ellc_copy_slot_across: the two contexts live in different libraries")
    st = dst_ctx._l.ellc_copy_slot_across(dst_ctx.h, int(dst_is_kf), dst, src_ctx.h, int(src_is_kf), src)
    if st != 0:
        raise EllcError("ellc_copy_slot_across -> %d: %s" % (st, dst_ctx._l.ellc_last_error(dst_ctx.h).decode()))


def concatenate_relative_pose(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    o = np.zeros(6, np.float32)
    _lib.lib().ellc_concatenate_relative_pose(_p(a), _p(b), _p(o))
    return o


def concatenate_origin_pose(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    o = np.zeros(6, np.float32)
    _lib.lib().ellc_concatenate_origin_pose(_p(a), _p(b), _p(o))
    return o


def se3_exp(pose):
    p = np.ascontiguousarray(pose, np.float32)
    T = np.zeros(16, np.float32)
    _lib.lib().ellc_se3_exp(_p(p), _p(T))
    return T.reshape(4, 4)


def se3_log(T):
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    p = np.zeros(6, np.float32)
    _lib.lib().ellc_se3_log(_p(T), _p(p))
    return p


def kl_divergence(p, q):
    p = np.ascontiguousarray(p, np.float32); q = np.ascontiguousarray(q, np.float32)
    return float(_lib.lib().ellc_kl_divergence(_p(p), _p(q), p.size))
