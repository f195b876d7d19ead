"""Known-answer tests that pin the oracle's image side and Gauss-Newton path (the reference has no tests or
fixtures: SURVEY.md §4/§8c). Every expectation is analytic or comes from an independent numpy evaluation."""
import os
import numpy as np
import pytest
from egomotion_with_local_loop_closures_amd import synth
from helpers import oracle_problem


def test_pyrdown_constant_and_impulse(oracle):
    img = np.full((48, 64), 77, np.uint8)
    assert np.all(oracle.pyr_down(img) == 77)
    # impulse response of the separable [1 4 6 4 1]/16 kernel with (sum+128)>>8 rounding
    img = np.zeros((33, 41), np.uint8)
    img[16, 20] = 255
    out = oracle.pyr_down(img)
    assert out.shape == (17, 21)                      # ceil rule
    k = np.array([1, 4, 6, 4, 1])
    exp = np.zeros((17, 21), np.int64)
    for dy in range(-1, 2):
        for dx in range(-1, 2):
            wy = k[2 + 2 * dy] if abs(2 * dy) <= 2 else 0
            wx = k[2 + 2 * dx] if abs(2 * dx) <= 2 else 0
            exp[8 + dy, 10 + dx] = (255 * wy * wx + 128) >> 8
    assert np.array_equal(out, exp)


def test_pyrdown_matches_numpy_reference(oracle):
    rng = np.random.default_rng(0)
    for (h, w) in ((30, 44), (31, 45), (5, 7)):
        img = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
        k = np.array([1, 4, 6, 4, 1], np.int64)
        pad = np.pad(img.astype(np.int64), 2, mode="reflect")            # reflect == BORDER_REFLECT_101
        hs = sum(k[i] * pad[:, i:i + w] for i in range(5))
        vs = sum(k[i] * hs[i:i + h, :] for i in range(5))
        exp = ((vs[::2, ::2] + 128) >> 8).astype(np.uint8)
        assert np.array_equal(oracle.pyr_down(img), exp)


def test_gradient_ramp_and_borders(oracle):
    x = np.arange(40, dtype=np.uint8)[None, :] * 3
    img = np.repeat(x, 20, axis=0)
    gx, gy = oracle.gradient(img)
    assert np.all(gx[:, 1:-1] == 3.0)                # interior: 0.5*(I[x+1]-I[x-1])
    assert np.all(gx[:, 0] == 3.0) and np.all(gx[:, -1] == 3.0)   # border: one-sided, no 0.5 (Q12)
    assert np.all(gy == 0)
    img2 = img.T.copy()
    gx2, gy2 = oracle.gradient(img2)
    assert np.all(gy2[1:-1, :] == 3.0) and np.all(gy2[0, :] == 3.0) and np.all(gx2 == 0)


def test_max_gradient_is_3x3_max_inside_and_raw_on_border(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(24, 32)).astype(np.uint8)
    gx, gy = oracle.gradient(img)
    mg, n = oracle.max_gradients(gx, gy)
    mag = np.sqrt(gx * gx + gy * gy)
    exp = mag.copy()
    for y in range(1, 23):
        for x in range(1, 31):
            exp[y, x] = max(mag[yy, xx] for yy in (y - 1, y, y + 1) for xx in (x - 1, x, x + 1)
                            if 1 <= yy <= 22) if True else 0
    # vertical pass only covers rows 1..h-2 (tmp is 0 elsewhere) and the horizontal pass reads tmp of the same row
    tmp = np.zeros_like(mag)
    tmp[1:-1, :] = np.maximum(np.maximum(mag[1:-1, :], mag[:-2, :]), mag[2:, :])
    exp = mag.copy()
    exp[1:-1, 1:-1] = np.maximum(np.maximum(tmp[1:-1, :-2], tmp[1:-1, 1:-1]), tmp[1:-1, 2:])
    assert np.array_equal(mg, exp)
    assert n == int((exp[1:-1, 1:-1] >= 5).sum())


def test_bilinear_tap_semantics(oracle):
    img = (np.arange(12, dtype=np.uint8).reshape(3, 4) * 10)
    xs = np.array([0.0, 1.5, 2.25, 3.0, 3.5, -0.5, 1.0, 1.0, 4.0], np.float32)
    ys = np.array([0.0, 0.5, 1.75, 2.0, 1.0, 1.0, -0.1, 2.5, 1.0], np.float32)
    out = oracle.tap_u8(img, xs, ys, check=1)
    assert out[0] == 0 and out[3] == 110                                   # exact pixel hits
    assert out[1] == pytest.approx(0.25 * (10 + 20 + 50 + 60))            # interior bilinear
    assert out[2] == pytest.approx((1 - .75) * (.75 * 60 + .25 * 70) + .75 * (.75 * 100 + .25 * 110))
    assert out[4] == pytest.approx(0.5 * 70)                               # x in (W-1, W): right taps are 0, not invalid (Q3)
    assert out[5] == -1 and out[6] == -1 and out[8] == -1                  # x<0 or y<0 or x>=W: all four OOB => -1
    assert out[7] == pytest.approx(0.5 * 90)                               # y in (H-1, H): bottom taps 0
    assert np.all(oracle.tap_u8(img, xs[[5, 6, 8]], ys[[5, 6, 8]], check=0) == 0)   # no sentinel without the check


def test_gn_identity_pose_has_zero_residual_and_update(oracle):
    """current == keyframe and pose 0 => r = 0 everywhere => b = 0 => delta = 0 (analytic fixed point)."""
    w, h, L = 96, 72, 3
    pair = synth.make_pair(w, h, seed=8)
    pair["cur_image"] = pair["kf_image"].copy()
    _, kf, cur, dm = oracle_problem(oracle, w, h, L, pair, max_iter=(4, 7, 9))
    st = oracle.GNStepper(kf, cur, dm.depth_pyr(), 0, np.zeros(6, np.float32), planes=True)
    r = st.step(0)
    pl = st.get_planes()
    # the f32 back-project / re-project round trip lands within a few ulp of the pixel centre, so r is ~1e-5, not 0
    assert np.abs(pl["residual"]).max() < 1e-4
    assert np.abs(r["delta"]).max() < 1e-7 and r["weighted"] < 0.01
    assert np.abs(r["pose"]).max() < 1e-7
    mask = kf.depth(0) > 0
    assert np.array_equal(pl["weight"][mask], np.full(mask.sum(), 1.0 / 16.0, np.float32))   # t = 0 => w = 1/CAMERA_PIXEL_NOISE_2
    ys, xs = np.nonzero(mask)
    assert np.abs(pl["warpedX"][mask] - xs).max() < 1e-4 and np.abs(pl["warpedY"][mask] - ys).max() < 1e-4
    assert np.allclose(r["H"], r["H"].T, rtol=1e-5)
    st.close()


def test_gn_jacobian_matches_independent_numpy(oracle):
    """J of PixelWisePyramid.cpp:296-320 re-derived in float64 numpy from the gradient planes."""
    w, h, L = 96, 72, 3
    pair = synth.make_pair(w, h, seed=9)
    _, kf, cur, dm = oracle_problem(oracle, w, h, L, pair, max_iter=(4, 7, 9))
    pose = np.array([0.003, -0.002, 0.001, 0.004, -0.003, 0.002], np.float32)
    level = 1
    st = oracle.GNStepper(kf, cur, dm.depth_pyr(), level, pose, planes=True)
    r = st.step(0)
    pl = st.get_planes()
    fx, fy, cx, cy = oracle.get_intrinsic(kf.cfg, level).astype(np.float64)
    Z = kf.depth(level).astype(np.float64)
    mask = (Z > 0) & (pl["warpedX"] >= 0)
    cur.update_level(level, False)
    gxp, gyp = cur.gradient(level)
    gx = oracle.tap_f32(gxp, pl["warpedX"][mask], pl["warpedY"][mask]).astype(np.float64)
    gy = oracle.tap_f32(gyp, pl["warpedX"][mask], pl["warpedY"][mask]).astype(np.float64)
    ys, xs = np.nonzero(mask)
    u = xs - cx; v = ys - cy; z = Z[mask]
    J = np.stack([gx * (-u * v / fy) + gy * (-(fy + v * v / fy)), gx * (fx + u * u / fx) + gy * (u * v / fx),
                  gx * (-fx * v / fy) + gy * (fy * u / fx), gx * fx / z, gy * fy / z, -(gx * u + gy * v) / z])
    got = np.stack([pl["J"][k][mask] for k in range(6)])
    assert np.allclose(got, J, rtol=2e-5, atol=2e-4)
    # H and b are the weighted sums of those rows
    wgt = pl["weight"][mask].astype(np.float64); res = pl["residual"][mask].astype(np.float64)
    assert np.allclose(r["Hd"], (J * wgt) @ J.T, rtol=1e-4)
    assert np.allclose(r["bd"], J @ (wgt * res), rtol=1e-3, atol=1e-6 * np.abs(r["bd"]).max() + 1e-3)
    st.close()


def test_gn_converges_to_synthetic_motion(oracle):
    w, h, L = 320, 240, 4
    pair = synth.make_pair(w, h, seed=1)
    _, kf, cur, dm = oracle_problem(oracle, w, h, L, pair)
    pose, iters, wgt = oracle.align(kf, cur, dm.depth_pyr())
    assert list(iters) == [4, 7, 9, 12]
    assert np.linalg.norm(pose - pair["xi_true"]) < 1.5e-3
    p64, _, _ = oracle.align(kf, cur, dm.depth_pyr(), sum_mode=1)
    assert np.linalg.norm(pose - p64) < 1e-6          # band-summed f32 vs f64-summed terms
    # threaded bands (CPU-baseline mode) give the same bits as serial bands
    pt, _, _ = oracle.align(kf, cur, dm.depth_pyr(), spawn_threads=True)
    assert np.array_equal(pt, pose)


def test_gn_early_exit_and_pose_bookkeeping(oracle):
    w, h, L = 160, 120, 3
    pair = synth.make_pair(w, h, seed=12, rot=0.002, trans=0.004)
    _, kf, cur, dm = oracle_problem(oracle, w, h, L, pair, early_exit=1, max_iter=(4, 7, 9))
    kf.set_pose(world=np.array([0.01, 0, 0, 0.1, 0, 0], np.float32))
    pose, iters, wgt = oracle.align(kf, cur, dm.depth_pyr())
    assert wgt < 1.0 and np.all(iters <= np.array([4, 7, 9])) and iters.sum() < 20
    o, wld = cur.pose()
    assert np.allclose(o, pose, atol=1e-7)                                 # keyframe origin pose is 0
    assert np.allclose(wld, oracle.concat_relative(pose, np.array([0.01, 0, 0, 0.1, 0, 0], np.float32)), atol=1e-7)


def test_golden_gn_small_regression(oracle):
    """tests/golden/gn_small.npz (written by tests/golden/make_golden.py from the oracle): pins the restatement."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gn_small.npz"))
    pair = dict(kf_image=g["kf_image"], cur_image=g["cur_image"], depth0=g["depth0"], var0=g["var0"], intrinsics=tuple(g["intrinsics"]))
    _, kf, cur, dm = oracle_problem(oracle, 64, 48, 3, pair, max_iter=tuple(g["max_iter"]))
    for level in (2, 1, 0):
        st = oracle.GNStepper(kf, cur, dm.depth_pyr(), level, g["L%d_pose_in" % level], planes=True)
        r = st.step(0)
        pl = st.get_planes()
        assert np.array_equal(pl["residual"], g["L%d_residual" % level])
        assert np.array_equal(pl["weight"], g["L%d_weight" % level])
        assert np.array_equal(pl["J"], g["L%d_J" % level])
        assert np.array_equal(r["H"], g["L%d_it0_H" % level]) and np.array_equal(r["pose"], g["L%d_it0_pose" % level])
        st.close()
    pose, iters, _ = oracle.align(kf, cur, dm.depth_pyr())
    assert np.array_equal(pose, g["final_pose"]) and np.array_equal(iters, g["iters"])
