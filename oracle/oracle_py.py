"""ctypes wrapper over oracle/libellc_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (egomotion_with_local_loop_closures_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libellc_oracle.so")
MAX_LEVELS = 8


class OrcConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("levels", C.c_int),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("max_iter", C.c_int * MAX_LEVELS), ("early_exit", C.c_int), ("num_pose_threads", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_frame_create.restype = C.c_void_p
        _lib.orc_depthpyr_create.restype = C.c_void_p
        _lib.orc_gn_begin.restype = C.c_void_p
        _lib.orc_dm_create.restype = C.c_void_p
        _lib.orc_dm_pyr.restype = C.c_void_p
        _lib.orc_align_timed.restype = C.c_double
        _lib.orc_align_batch_timed.restype = C.c_double
        _lib.orc_dm_make_inv_depth_one.restype = C.c_float
        _lib.orc_dm_seeds.restype = C.c_float
        _lib.orc_dm_line_stereo.restype = C.c_float
        _lib.orc_frame_get_rescale.restype = C.c_float
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def make_config(width, height, levels, fx, fy, cx, cy, max_iter=(4, 7, 9, 12), early_exit=1, threads=3):
    c = OrcConfig()
    c.width, c.height, c.levels = width, height, levels
    c.fx, c.fy, c.cx, c.cy = fx, fy, cx, cy
    mi = list(max_iter) + [12] * (MAX_LEVELS - len(max_iter))
    for i in range(MAX_LEVELS):
        c.max_iter[i] = mi[i]
    c.early_exit = early_exit
    c.num_pose_threads = threads
    return c


# ---------------------------------------------------------------- algebra
def se3_exp(pose):
    pose = np.ascontiguousarray(pose, np.float32)
    T = np.zeros(16, np.float32)
    lib().orc_se3_exp(_p(pose), _p(T))
    return T.reshape(4, 4)


def se3_log(T):
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    p = np.zeros(6, np.float32)
    lib().orc_se3_log(_p(T), _p(p))
    return p


def se3_exp_d(pose):
    pose = np.ascontiguousarray(pose, np.float64)
    T = np.zeros(16, np.float64)
    lib().orc_se3_exp_d(_p(pose), _p(T))
    return T.reshape(4, 4)


def se3_log_d(T):
    T = np.ascontiguousarray(T, np.float64).reshape(16)
    p = np.zeros(6, np.float64)
    lib().orc_se3_log_d(_p(T), _p(p))
    return p


def concat_relative(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    o = np.zeros(6, np.float32)
    lib().orc_concat_relative(_p(a), _p(b), _p(o))
    return o


def concat_origin(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    o = np.zeros(6, np.float32)
    lib().orc_concat_origin(_p(a), _p(b), _p(o))
    return o


def lu_inverse(A):
    A = np.ascontiguousarray(A, np.float32)
    n = A.shape[0]
    out = np.zeros((n, n), np.float32)
    ok = lib().orc_lu_inverse(_p(A), n, _p(out))
    return ok, out


def get_intrinsic(cfg, level):
    o = np.zeros(4, np.float32)
    lib().orc_get_intrinsic(C.byref(cfg), level, _p(o))
    return o


def kmats(cfg):
    K = np.zeros(9, np.float32); Ki = np.zeros(9, np.float32)
    lib().orc_kmats(C.byref(cfg), _p(K), _p(Ki))
    return K.reshape(3, 3), Ki.reshape(3, 3)


# ---------------------------------------------------------------- image side
def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down(_p(img), w, h, _p(out))
    return out


def gradient(img, rows=None, cols=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    rows = h if rows is None else rows
    cols = w if cols is None else cols
    gx = np.zeros((rows, cols), np.float32); gy = np.zeros((rows, cols), np.float32)
    lib().orc_gradient(_p(img), w, h, rows, cols, _p(gx), _p(gy))
    return gx, gy


def max_gradients(gx, gy):
    gx = np.ascontiguousarray(gx, np.float32); gy = np.ascontiguousarray(gy, np.float32)
    h, w = gx.shape
    out = np.zeros((h, w), np.float32)
    n = C.c_int(0)
    lib().orc_max_gradients(_p(gx), _p(gy), w, h, _p(out), C.byref(n))
    return out, n.value


def tap_u8(img, xs, ys, check=1, rows=None, cols=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    xs = np.ascontiguousarray(xs, np.float32); ys = np.ascontiguousarray(ys, np.float32)
    out = np.zeros(xs.shape, np.float32)
    lib().orc_tap_u8(_p(img), w, h, h if rows is None else rows, w if cols is None else cols, _p(xs), _p(ys), xs.size, check, _p(out))
    return out


def tap_f32(img, xs, ys):
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    xs = np.ascontiguousarray(xs, np.float32); ys = np.ascontiguousarray(ys, np.float32)
    out = np.zeros(xs.shape, np.float32)
    lib().orc_tap_f32(_p(img), w, h, _p(xs), _p(ys), xs.size, _p(out))
    return out


# ---------------------------------------------------------------- handles
class Frame:
    def __init__(self, cfg, gray, frame_id=1):
        self.cfg = cfg
        gray = np.ascontiguousarray(gray, np.uint8)
        assert gray.shape == (cfg.height, cfg.width)
        self.h = C.c_void_p(lib().orc_frame_create(C.byref(cfg), _p(gray), frame_id))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_frame_destroy(self.h)
            self.h = None

    def level_dims(self, level):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().orc_frame_level_dims(self.h, level, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return a.value, b.value, c.value, d.value  # stored_w, stored_h, cols, rows

    def image(self, level):
        sw, sh, _, _ = self.level_dims(level)
        out = np.zeros((sh, sw), np.uint8)
        lib().orc_frame_get_image(self.h, level, _p(out))
        return out

    def set_depth(self, level, depth):
        depth = np.ascontiguousarray(depth, np.float32)
        lib().orc_frame_set_depth(self.h, level, _p(depth))

    def depth(self, level):
        _, _, c, r = self.level_dims(level)
        out = np.zeros((r, c), np.float32)
        lib().orc_frame_get_depth(self.h, level, _p(out))
        return out

    def set_weights(self, level, w, count=1):
        w = np.ascontiguousarray(w, np.float32)
        lib().orc_frame_set_weights(self.h, level, _p(w), count)

    def weights(self, level):
        _, _, c, r = self.level_dims(level)
        out = np.zeros((r, c), np.float32)
        n = C.c_int(0)
        lib().orc_frame_get_weights(self.h, level, _p(out), C.byref(n))
        return out, n.value

    def finalise_weights(self):
        lib().orc_frame_finalise_weights(self.h)

    def max_gradient(self):
        out = np.zeros((self.cfg.height, self.cfg.width), np.float32)
        n = C.c_int(0)
        lib().orc_frame_get_max_gradient(self.h, _p(out), C.byref(n))
        return out, n.value

    def set_pose(self, origin=None, world=None):
        o = None if origin is None else np.ascontiguousarray(origin, np.float32)
        w = None if world is None else np.ascontiguousarray(world, np.float32)
        lib().orc_frame_set_pose(self.h, _p(o), _p(w))

    def calc_se3(self, other):
        """calculateSE3poseOtherWrtThis(other) (Frame.cpp:376-413); returns the matrices it leaves in this frame."""
        lib().orc_frame_calc_se3(self.h, other.h)
        m = np.zeros(44, np.float32)
        lib().orc_frame_get_se3(self.h, _p(m))
        return {"OtherWrtThis": m[:16].reshape(4, 4).copy(), "ThisWrtOther": m[16:32].reshape(4, 4).copy(), "K_r": m[32:41].reshape(3, 3).copy(),
                "K_t": m[41:44].copy()}

    def kinv(self):
        """ORIG_FX_INV, ORIG_FY_INV, ORIG_CX_INV, ORIG_CY_INV (EigenInitialization.cpp:20-34)"""
        o = np.zeros(4, np.float32)
        lib().orc_kmats_inv(self.h, _p(o))
        return o

    def pose(self):
        o = np.zeros(6, np.float32); w = np.zeros(6, np.float32)
        lib().orc_frame_get_pose(self.h, _p(o), _p(w))
        return o, w

    def rescale_factor(self):
        return lib().orc_frame_get_rescale(self.h)

    def set_early_exit(self, e):
        lib().orc_frame_set_early_exit(self.h, int(e))

    def set_max_iter(self, mi):
        a = (C.c_int * MAX_LEVELS)(*(list(mi) + [12] * (MAX_LEVELS - len(mi))))
        lib().orc_frame_set_max_iter(self.h, a)

    def update_level(self, level, is_prev=True):
        lib().orc_frame_update_level(self.h, level, int(is_prev))

    def gradient(self, level):
        _, _, c, r = self.level_dims(level)
        gx = np.zeros((r, c), np.float32); gy = np.zeros((r, c), np.float32)
        lib().orc_frame_get_gradient(self.h, _p(gx), _p(gy))
        return gx, gy

    def mask(self, level):
        _, _, c, r = self.level_dims(level)
        m = np.zeros((r, c), np.uint8)
        n = lib().orc_frame_get_mask(self.h, _p(m))
        return m, n


class DepthPyr:
    def __init__(self, cfg, handle=None):
        self.cfg = cfg
        self.owned = handle is None
        self.h = C.c_void_p(lib().orc_depthpyr_create(C.byref(cfg))) if handle is None else handle

    def __del__(self):
        if getattr(self, "h", None) and self.owned:
            lib().orc_depthpyr_destroy(self.h)
            self.h = None

    def set_var(self, level, var):
        var = np.ascontiguousarray(var, np.float32)
        lib().orc_depthpyr_set_var(self.h, level, _p(var))


class GNStepper:
    """One pyramid level of PixelWisePyramid; step() = one GN iteration."""

    def __init__(self, kf, cur, dpyr, level, pose, sum_mode=0, n_threads=3, planes=False):
        self.kf, self.cur, self.dpyr = kf, cur, dpyr
        pose = np.ascontiguousarray(pose, np.float32)
        self.level = level
        self.planes = planes
        self.h = C.c_void_p(lib().orc_gn_begin(kf.h, cur.h, dpyr.h, level, _p(pose), sum_mode, n_threads, int(planes)))
        _, _, self.cols, self.rows = kf.level_dims(level)

    def step(self, mode=0, it=0):
        H = np.zeros((6, 6), np.float32); b = np.zeros(6, np.float32); Hi = np.zeros((6, 6), np.float32)
        d = np.zeros(6, np.float32); p = np.zeros(6, np.float32); w = C.c_float(0)
        Hd = np.zeros((6, 6), np.float64); bd = np.zeros(6, np.float64)
        lib().orc_gn_step(self.h, mode, it, _p(H), _p(b), _p(Hi), _p(d), _p(p), C.byref(w), _p(Hd), _p(bd))
        return dict(H=H, b=b, Hinv=Hi, delta=d, pose=p, weighted=w.value, Hd=Hd, bd=bd)

    def get_planes(self):
        n = self.rows * self.cols
        r = np.zeros((self.rows, self.cols), np.float32); w = np.zeros_like(r); wx = np.zeros_like(r); wy = np.zeros_like(r)
        J = np.zeros((6, self.rows, self.cols), np.float32)
        lib().orc_gn_planes(self.h, _p(r), _p(w), _p(wx), _p(wy), _p(J))
        return dict(residual=r, weight=w, warpedX=wx, warpedY=wy, J=J)

    def get_sd(self):
        n = self.rows * self.cols
        sd = np.zeros((6, n), np.float32); wsd = np.zeros((6, n), np.float32)
        lib().orc_gn_sd(self.h, _p(sd), _p(wsd))
        return sd, wsd

    def save_weights(self):
        lib().orc_gn_save_weights(self.h)

    def close(self):
        if self.h:
            lib().orc_gn_end(self.h)
            self.h = None

    def __del__(self):
        self.close()


def align(kf, cur, dpyr, init_pose=None, loop_closure=False, save_weights=False, spawn_threads=False, sum_mode=0, n_threads=3):
    ip = np.zeros(6, np.float32) if init_pose is None else np.ascontiguousarray(init_pose, np.float32)
    pose = np.zeros(6, np.float32)
    iters = np.zeros(MAX_LEVELS, np.int32)
    w = C.c_float(0)
    flags = (1 if loop_closure else 0) | (2 if save_weights else 0) | (4 if spawn_threads else 0)
    lib().orc_align(kf.h, cur.h, dpyr.h, _p(ip), flags, sum_mode, n_threads, _p(pose), _p(iters), C.byref(w))
    return pose, iters[:kf.cfg.levels].copy(), w.value


def align_timed(kf, cur, dpyr, init_pose=None, loop_closure=False, spawn_threads=True, n_threads=3, reps=1, pool=False):
    """CPU baseline timing. spawn_threads: bands on std::threads created / joined per iteration (the reference's way);
    pool: a persistent worker pool instead; neither: bands one after the other on the calling thread."""
    ip = np.zeros(6, np.float32) if init_pose is None else np.ascontiguousarray(init_pose, np.float32)
    its = C.c_longlong(0)
    flags = (1 if loop_closure else 0) | (8 if pool else (4 if spawn_threads else 0))
    sec = lib().orc_align_timed(kf.h, cur.h, dpyr.h, _p(ip), flags, n_threads, reps, C.byref(its))
    return sec, its.value


def align_batch_timed(problems, n_outer, loop_closure=False, spawn_threads=False, n_threads=3, reps=1):
    """CPU baseline, batch-parallel (orc_align_batch_timed): problems = [(kf, cur, dpyr), ...], each with objects of its own;
    alignment i runs on host thread i % n_outer. Returns (seconds, GN iterations executed)."""
    n = len(problems)
    kfs = (C.c_void_p * n)(*[p[0].h for p in problems])
    curs = (C.c_void_p * n)(*[p[1].h for p in problems])
    dps = (C.c_void_p * n)(*[p[2].h for p in problems])
    its = C.c_longlong(0)
    flags = (1 if loop_closure else 0) | (4 if spawn_threads else 0)
    sec = lib().orc_align_batch_timed(kfs, curs, dps, n, int(n_outer), flags, n_threads, reps, C.byref(its))
    return sec, its.value


def hardware_threads():
    return lib().orc_hardware_threads()


HYP_FIELDS = ("invDepth", "invDepthSmoothed", "variance", "varianceSmoothed", "validity", "blacklisted", "valid")


class DepthMap:
    def __init__(self, cfg):
        self.cfg = cfg
        self.h = C.c_void_p(lib().orc_dm_create(C.byref(cfg)))
        self.n = cfg.width * cfg.height
        self._kf = None
        self._cur = None

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_dm_destroy(self.h)
            self.h = None

    def set_state(self, st):
        a = [np.ascontiguousarray(st["invDepth"], np.float32), np.ascontiguousarray(st["invDepthSmoothed"], np.float32),
             np.ascontiguousarray(st["variance"], np.float32), np.ascontiguousarray(st["varianceSmoothed"], np.float32),
             np.ascontiguousarray(st["validity"], np.int32), np.ascontiguousarray(st["blacklisted"], np.int32),
             np.ascontiguousarray(st["valid"], np.uint8)]
        lib().orc_dm_set_state(self.h, *[_p(x) for x in a])

    def get_state(self):
        shp = (self.cfg.height, self.cfg.width)
        a = [np.zeros(shp, np.float32), np.zeros(shp, np.float32), np.zeros(shp, np.float32), np.zeros(shp, np.float32),
             np.zeros(shp, np.int32), np.zeros(shp, np.int32), np.zeros(shp, np.uint8)]
        lib().orc_dm_get_state(self.h, *[_p(x) for x in a])
        return dict(zip(HYP_FIELDS, a))

    def set_keyframe(self, f):
        self._kf = f
        lib().orc_dm_set_keyframe(self.h, f.h)

    def set_current(self, f):
        self._cur = f
        lib().orc_dm_set_current(self.h, f.h)

    def propagate(self, newkf):
        lib().orc_dm_propagate(self.h, newkf.h)

    def observe(self):
        lib().orc_dm_observe(self.h)

    def fill_holes(self):
        lib().orc_dm_fill_holes(self.h)

    def regularize(self, remove_occlusions=False):
        lib().orc_dm_regularize(self.h, int(remove_occlusions))

    def make_inv_depth_one(self):
        return lib().orc_dm_make_inv_depth_one(self.h)

    def update_depth_image(self):
        lib().orc_dm_update_depth_image(self.h)

    def create_keyframe(self, newkf):
        lib().orc_dm_create_keyframe(self.h, newkf.h)
        self._kf = newkf

    def seeds(self):
        return lib().orc_dm_seeds(self.h)

    def pyr_level(self, level):
        shp = (self.cfg.height >> level, self.cfg.width >> level)
        d = np.zeros(shp, np.float32); v = np.zeros(shp, np.float32)
        lib().orc_dm_get_pyr(self.h, level, _p(d), _p(v))
        return d, v

    def integral(self):
        out = np.zeros((self.cfg.height, self.cfg.width), np.int32)
        lib().orc_dm_get_integral(self.h, _p(out))
        return out

    def depth_pyr(self):
        dp = DepthPyr(self.cfg, handle=C.c_void_p(lib().orc_dm_pyr(self.h)))
        dp._map = self   # the pyramid lives inside the map: keep the map alive as long as the view
        return dp

    def set_pyr0(self, deptharr0, vararr0):
        a = np.ascontiguousarray(deptharr0, np.float32); b = np.ascontiguousarray(vararr0, np.float32)
        lib().orc_dm_set_pyr0(self.h, _p(a), _p(b))

    def build_inv_var_depth(self):
        lib().orc_dm_build_inv_var_depth(self.h)

    def map_depth_to_keyframe(self):
        lib().orc_dm_map_depth_to_keyframe(self.h)

    def line_stereo(self, u, v, epxn, epyn, min_id, prior, max_id):
        out = np.zeros(3, np.float32)
        e = lib().orc_dm_line_stereo(self.h, C.c_float(u), C.c_float(v), C.c_float(epxn), C.c_float(epyn), C.c_float(min_id),
                                     C.c_float(prior), C.c_float(max_id), _p(out))
        return e, out

    def check_epl(self, x, y):
        ep = np.zeros(2, np.float32)
        ok = lib().orc_dm_check_epl(self.h, x, y, _p(ep))
        return ok, ep
