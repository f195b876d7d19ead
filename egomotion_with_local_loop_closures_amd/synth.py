"""Seeded synthetic inputs for the alignment / depth hot path (SURVEY.md §8(d) recipe).

No reference code or data is involved: textures are value noise, the scene is a smooth inverse-depth
field, the "current" image is the keyframe texture re-rendered under a known SE(3) motion, and the
keyframe's semi-dense depth map is the true field plus measurement noise on the high-gradient pixels.
"""
import numpy as np


def _bilinear_upsample(grid, h, w):
    gh, gw = grid.shape
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.clip(np.floor(ys).astype(int), 0, gh - 2)
    x0 = np.clip(np.floor(xs).astype(int), 0, gw - 2)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = grid[y0][:, x0]
    b = grid[y0][:, x0 + 1]
    c = grid[y0 + 1][:, x0]
    d = grid[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def value_noise_texture(w, h, rng, cells=(64, 32, 16, 8), amps=(80, 48, 32, 20)):
    """u8 texture: sum of octaves of bilinear value noise + 128."""
    img = np.full((h, w), 128.0)
    for cell, amp in zip(cells, amps):
        gh, gw = h // cell + 2, w // cell + 2
        img += amp * (_bilinear_upsample(rng.random((gh, gw)), h, w) - 0.5) * 2.0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def smooth_field(w, h, rng, cell=128, lo=0.6, hi=1.4):
    gh, gw = h // cell + 2, w // cell + 2
    return lo + (hi - lo) * _bilinear_upsample(rng.random((gh, gw)), h, w)


def se3_exp(xi):
    """closed-form exp: xi = [w, v] -> 4x4 (float64)."""
    xi = np.asarray(xi, np.float64)
    w, v = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-8:
        A, B, Cc = 1.0, 0.5, 1.0 / 6
    else:
        A = np.sin(th) / th
        B = (1 - np.cos(th)) / th ** 2
        Cc = (th - np.sin(th)) / th ** 3
    R = np.eye(3) + A * W + B * W @ W
    V = np.eye(3) + B * W + Cc * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


def _bilinear_sample(img, x, y):
    h, w = img.shape
    x = np.clip(x, 0, w - 1.001)
    y = np.clip(y, 0, h - 1.001)
    x0 = np.floor(x).astype(int)
    y0 = np.floor(y).astype(int)
    fx = x - x0
    fy = y - y0
    a = img[y0, x0]; b = img[y0, x0 + 1]; c = img[y0 + 1, x0]; d = img[y0 + 1, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def render_current(tex, idepth_true, T, fx, fy, cx, cy, iters=8, with_depth=False):
    """Image seen after the camera motion T (P' = R P + t): inverse warp by fixed-point iteration.
    with_depth: also return the inverse depth of the surface as seen from the new view (f64)."""
    h, w = tex.shape
    texf = tex.astype(np.float64)
    uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    x, y = uu.copy(), vv.copy()
    R, t = T[:3, :3], T[:3, 3]
    for _ in range(iters):
        Z = 1.0 / _bilinear_sample(idepth_true, x, y)
        X = (x - cx) * Z / fx
        Y = (y - cy) * Z / fy
        Xp = R[0, 0] * X + R[0, 1] * Y + R[0, 2] * Z + t[0]
        Yp = R[1, 0] * X + R[1, 1] * Y + R[1, 2] * Z + t[1]
        Zp = R[2, 0] * X + R[2, 1] * Y + R[2, 2] * Z + t[2]
        uw = fx * Xp / Zp + cx
        vw = fy * Yp / Zp + cy
        x -= (uw - uu)
        y -= (vw - vv)
    out = _bilinear_sample(texf, x, y)
    img = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    if not with_depth:
        return img
    Z = 1.0 / _bilinear_sample(idepth_true, x, y)
    X = (x - cx) * Z / fx
    Y = (y - cy) * Z / fy
    Zp = R[2, 0] * X + R[2, 1] * Y + R[2, 2] * Z + t[2]
    return img, 1.0 / Zp


def max_abs_gradient(img):
    """numpy twin of the keyframe's 3x3-max gradient magnitude (selection only; parity is tested elsewhere)."""
    f = img.astype(np.float32)
    gx = np.zeros_like(f); gy = np.zeros_like(f)
    gx[:, 1:-1] = 0.5 * (f[:, 2:] - f[:, :-2]); gx[:, 0] = f[:, 1] - f[:, 0]; gx[:, -1] = f[:, -1] - f[:, -2]
    gy[1:-1, :] = 0.5 * (f[2:, :] - f[:-2, :]); gy[0, :] = f[1, :] - f[0, :]; gy[-1, :] = f[-1, :] - f[-2, :]
    g = np.sqrt(gx * gx + gy * gy)
    t = g.copy()
    t[1:-1, :] = np.maximum(np.maximum(g[1:-1, :], g[:-2, :]), g[2:, :])
    o = g.copy()
    o[1:-1, 1:-1] = np.maximum(np.maximum(t[1:-1, :-2], t[1:-1, 1:-1]), t[1:-1, 2:])
    return o


def default_intrinsics(w, h):
    # same fx/W ratio as ExternVariable.h:53-59 (410.6 / 480)
    f = np.float32(0.855 * w)
    return float(f), float(f), float(w / 2.0), float(h / 2.0)


def make_pair(w, h, seed, dense=False, rot=0.01, trans=0.02, depth_noise=0.02, border=3):
    """One keyframe<->frame alignment problem.

    Returns dict: kf_image u8, cur_image u8, depth0 f32 (0 = no hypothesis), var0 f32 (-1 = none),
    idepth_true f64, xi_true (6,), intrinsics.
    """
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = default_intrinsics(w, h)
    tex = value_noise_texture(w, h, rng)
    idepth_true = smooth_field(w, h, rng)
    d = rng.normal(size=6)
    xi = np.concatenate([rot * d[:3] / np.linalg.norm(d[:3]), trans * d[3:] / np.linalg.norm(d[3:])])
    T = se3_exp(xi)
    cur = render_current(tex, idepth_true, T, fx, fy, cx, cy)
    idepth_meas = idepth_true * (1.0 + depth_noise * rng.normal(size=(h, w)))
    var = (0.125 * rng.uniform(0.5, 1.5, size=(h, w)) * 0.1).astype(np.float32)
    if dense:
        valid = np.ones((h, w), bool)
    else:
        valid = max_abs_gradient(tex) >= 5.0
    valid[:border, :] = False; valid[-border:, :] = False; valid[:, :border] = False; valid[:, -border:] = False
    depth0 = np.where(valid, 1.0 / idepth_meas, 0.0).astype(np.float32)
    var0 = np.where(valid, var, -1.0).astype(np.float32)
    return dict(kf_image=tex, cur_image=cur, depth0=depth0, var0=var0, idepth_true=idepth_true, xi_true=xi.astype(np.float32),
                intrinsics=(fx, fy, cx, cy), valid=valid)


def make_loop_closure_batch(w, h, B, seed, dense=False, rot=0.01, trans=0.02):
    """B distinct keyframes (textures, depth maps) each paired with its own view of the scene.

    The reference's loop-closure batch aligns B different keyframes against one current frame
    (GlobalOptimize.cpp:566); synthetic keyframes of *different* scenes cannot share one current
    image, so each alignment carries its own current frame — same work per alignment.
    """
    return [make_pair(w, h, seed + 7919 * b, dense=dense, rot=rot, trans=trans) for b in range(B)]


def make_shared_frame_batch(w, h, B, seed, dense=False, rot=0.01, trans=0.02, depth_noise=0.02, border=3):
    """The loop-closure batch in the reference's shape (GlobalOptimize.cpp:566): B different keyframes of ONE scene, each
    with its own semi-dense depth map, all aligned against ONE current frame. Every view is the scene texture re-rendered
    under its own known camera motion; alignment b's true pose is that of the current view relative to keyframe b.
    Returns a list of B dicts with make_pair's keys; every cur_image is the same array."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = default_intrinsics(w, h)
    tex = value_noise_texture(w, h, rng)
    idepth0 = smooth_field(w, h, rng)

    def motion(scale_r, scale_t):
        d = rng.normal(size=6)
        return np.concatenate([scale_r * d[:3] / np.linalg.norm(d[:3]), scale_t * d[3:] / np.linalg.norm(d[3:])])
    T_cur = se3_exp(motion(rot, trans))
    cur = render_current(tex, idepth0, T_cur, fx, fy, cx, cy, iters=6)
    out = []
    for b in range(B):
        T_kf = se3_exp(motion(rot, trans))
        kf_img, idepth_kf = render_current(tex, idepth0, T_kf, fx, fy, cx, cy, iters=6, with_depth=True)
        T_rel = T_cur @ np.linalg.inv(T_kf)          # keyframe b -> current view
        idepth_meas = idepth_kf * (1.0 + depth_noise * rng.normal(size=(h, w)))
        var = (0.125 * rng.uniform(0.5, 1.5, size=(h, w)) * 0.1).astype(np.float32)
        valid = np.ones((h, w), bool) if dense else (max_abs_gradient(kf_img) >= 5.0)
        valid[:border, :] = False; valid[-border:, :] = False; valid[:, :border] = False; valid[:, -border:] = False
        out.append(dict(kf_image=kf_img, cur_image=cur, depth0=np.where(valid, 1.0 / idepth_meas, 0.0).astype(np.float32),
                        var0=np.where(valid, var, -1.0).astype(np.float32), idepth_true=idepth_kf, xi_true=se3_log(T_rel).astype(np.float32),
                        intrinsics=(fx, fy, cx, cy), valid=valid))
    return out


def se3_log(T):
    """closed-form log of a 4x4 rigid transform (rotation angle < pi) -> [w, v] (float64)."""
    R, t = T[:3, :3], T[:3, 3]
    c = np.clip(0.5 * (np.trace(R) - 1.0), -1.0, 1.0)
    th = np.arccos(c)
    if th < 1e-8:
        w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    else:
        w = th / (2.0 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    th2 = float(w @ w)
    if th2 < 1e-12:
        Vinv = np.eye(3) - 0.5 * W + (1.0 / 12.0) * W @ W
    else:
        th = np.sqrt(th2)
        Vinv = np.eye(3) - 0.5 * W + (1.0 / th2) * (1.0 - (th * np.sin(th)) / (2.0 * (1.0 - np.cos(th)))) * W @ W
    return np.concatenate([w, Vinv @ t])


def make_depth_state(w, h, seed, kf_image, idepth_true=None, fill=0.9):
    """A plausible semi-dense hypothesis map (SoA dict) for the depth-map kernels."""
    rng = np.random.default_rng(seed)
    if idepth_true is None:
        idepth_true = smooth_field(w, h, rng)
    mg = max_abs_gradient(kf_image)
    valid = (mg >= 5.0) & (rng.random((h, w)) < fill)
    valid[:3, :] = False; valid[-3:, :] = False; valid[:, :3] = False; valid[:, -3:] = False
    idm = (idepth_true * (1.0 + 0.03 * rng.normal(size=(h, w)))).astype(np.float32)
    var = (0.01 * rng.uniform(0.5, 2.0, size=(h, w))).astype(np.float32)
    outlier = rng.random((h, w)) < 0.03          # gross outliers: exercised by the regulariser / occlusion logic
    idm = np.where(outlier, idm * rng.uniform(1.6, 2.5, size=(h, w)), idm).astype(np.float32)
    st = dict(invDepth=np.where(valid, idm, 0).astype(np.float32),
              invDepthSmoothed=np.where(valid, idm, 0).astype(np.float32),
              variance=np.where(valid, var, 0).astype(np.float32),
              varianceSmoothed=np.where(valid, var, 0).astype(np.float32),
              validity=np.where(valid, rng.integers(0, 60, size=(h, w)), 0).astype(np.int32),
              blacklisted=np.where(rng.random((h, w)) < 0.02, -rng.integers(1, 4, size=(h, w)), 0).astype(np.int32),
              valid=valid.astype(np.uint8))
    return st
